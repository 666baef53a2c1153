"""What the host offers the CPU baseline of bench.py: the CPUs this process may really use (affinity and the cgroup's
quota -- os.cpu_count() counts the machine's, not the container's) and whether the reference's CPU libraries
(FFTW3, Eigen3, TBB, essentia, MKL: reference CMakeLists.txt:34-36,42-54) are installed."""
import ctypes.util
import glob
import os


def cpu_budget():
    out = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpus": None}
    try:                                                   # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        out["cgroup_cpus"] = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_cpus"] = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    usable = out["affinity"]
    if out["cgroup_cpus"]:
        usable = min(usable, max(1, int(out["cgroup_cpus"] + 0.5)))
    out["usable"] = usable
    try:
        out["model"] = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        out["model"] = None
    return out


def reference_cpu_libraries():
    libs = {name: ctypes.util.find_library(name) for name in ("fftw3", "fftw3f", "tbb", "essentia", "mkl_rt")}
    headers = {}
    for name, pats in {"fftw3.h": ["/usr/include/fftw3.h", "/usr/local/include/fftw3.h", "/opt/*/include/fftw3.h"],
                       "Eigen/Core": ["/usr/include/eigen3/Eigen/Core", "/usr/local/include/eigen3/Eigen/Core",
                                      "/usr/include/Eigen/Core", "/opt/*/include/eigen3/Eigen/Core"],
                       "tbb/tbb.h": ["/usr/include/tbb/tbb.h", "/usr/include/oneapi/tbb.h", "/usr/local/include/tbb/tbb.h"],
                       "essentia/essentia.h": ["/usr/include/essentia/essentia.h", "/usr/local/include/essentia/essentia.h"]}.items():
        hit = [h for p in pats for h in glob.glob(p)]
        headers[name] = hit[0] if hit else None
    found = sorted([k for k, v in libs.items() if v] + [k for k, v in headers.items() if v])
    return {"libraries": libs, "headers": headers, "found": found,
            "fftw_eigen_tbb_variant": "not built: " + ("none of FFTW3 / Eigen3 / TBB is installed on this host" if not found else
                                                        "found " + ", ".join(found) + " but not the complete set the reference links")}
