// reg_probe.cpp -- can the GPU copy straight out of the page cache?  mmap a tmpfs file, hipHostRegister the mapping,
// hipMemcpy from it; against pread into a pinned buffer + hipMemcpy.  (diagnostic; build: hipcc tools/reg_probe.cpp -o tools/reg_probe.bin)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int nf = argc > 1 ? atoi(argv[1]) : 256, nt = argc > 2 ? atoi(argv[2]) : 16;
    const size_t sz = 2646044;
    const std::string dir = "/dev/shm/regp";
    (void)!system(("mkdir -p " + dir).c_str());
    std::vector<char> buf(sz, 3);
    for (int i = 0; i < nf; ++i) {
        int fd = open((dir + "/f" + std::to_string(i)).c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        (void)!write(fd, buf.data(), sz);
        close(fd);
    }
    char *dev = nullptr, *pin = nullptr;
    if (hipMalloc(&dev, nf * sz) != hipSuccess || hipHostMalloc(&pin, nf * sz, hipHostMallocDefault) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        // A: pread into pinned memory (team), one copy
        double t0 = now_ms();
        {
            std::atomic<int> next{0};
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&] {
                    for (int i; (i = next.fetch_add(1)) < nf;) {
                        int fd = open((dir + "/f" + std::to_string(i)).c_str(), O_RDONLY);
                        size_t done = 0;
                        while (done < sz) {
                            ssize_t r = pread(fd, pin + i * sz + done, sz - done, done);
                            if (r <= 0) break;
                            done += r;
                        }
                        close(fd);
                    }
                });
            for (auto &t : th) t.join();
        }
        double t1 = now_ms();
        (void)hipMemcpy(dev, pin, nf * sz, hipMemcpyHostToDevice);
        double t2 = now_ms();
        // B: register the mappings (team), copy each from the page cache, unregister
        std::vector<void *> maps(nf, nullptr);
        std::atomic<int> failed{0};
        {
            std::atomic<int> next{0};
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&] {
                    (void)hipSetDevice(0);
                    for (int i; (i = next.fetch_add(1)) < nf;) {
                        int fd = open((dir + "/f" + std::to_string(i)).c_str(), O_RDONLY);
                        void *m = mmap(nullptr, sz, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
                        close(fd);
                        if (m == MAP_FAILED || hipHostRegister(m, sz, hipHostRegisterDefault) != hipSuccess) {
                            failed++;
                            if (m != MAP_FAILED) munmap(m, sz);
                            (void)hipGetLastError();
                            continue;
                        }
                        maps[i] = m;
                    }
                });
            for (auto &t : th) t.join();
        }
        double t3 = now_ms();
        for (int i = 0; i < nf; ++i)
            if (maps[i]) (void)hipMemcpyAsync(dev + i * sz, maps[i], sz, hipMemcpyHostToDevice, nullptr);
        (void)hipDeviceSynchronize();
        double t4 = now_ms();
        for (int i = 0; i < nf; ++i)
            if (maps[i]) {
                (void)hipHostUnregister(maps[i]);
                munmap(maps[i], sz);
            }
        double t5 = now_ms();
        printf("%d files, %.1f MB, %d threads: pread->pinned %.1f ms + copy %.1f ms | register %.1f ms (failed %d) + copies %.1f ms + unregister %.1f ms\n",
               nf, nf * sz / 1e6, nt, t1 - t0, t2 - t1, t3 - t2, failed.load(), t4 - t3, t5 - t4);
    }
    (void)!system(("rm -rf " + dir).c_str());
    return 0;
}
