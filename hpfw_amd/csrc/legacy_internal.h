// legacy_internal.h -- the two halves of ParallelCollector::prepare (parallel_collector.h:48-52) exported by
// libhpfw_gpu.so for the multi-GPU host library (multi.cpp): preprocess' per-file loop (:85-105) and
// collect_fingerprints (:115-137), with the filter learning in between left to the caller, which sums accum_cov over
// the devices first.  Not part of the public C-ABI (include/hpfw_gpu.h declares par_collector_prepare, the whole).
#pragma once
#include "../../include/hpfw_gpu.h"

extern "C" {
typedef struct hpfw_prepare_job hpfw_prepare_job;
// a collector on `device` with `cache` loaded (filters.cereal, accum_cov.cereal; NULL / "" = "cache/")
hpfw_legacy_collector *hpfw_internal_collector_on_device(int device, const char *cache);
hpfw_gpu *hpfw_internal_collector_gpu(hpfw_legacy_collector *c);
int hpfw_internal_collector_set_filters(hpfw_legacy_collector *c, const float *filters_colmajor);
// spectrograms of the files, accum_cov += their covariances (learn != 0), spectrograms cached; NULL on host failure
hpfw_prepare_job *hpfw_internal_prepare_accumulate(hpfw_legacy_collector *c, const char **filenames, int n, int learn);
int64_t hpfw_internal_prepare_used(const hpfw_prepare_job *job); // files whose covariance was added
// hashprints under the installed filters; consumes the job
FilenameHashprintPair *hpfw_internal_prepare_finish(hpfw_legacy_collector *c, hpfw_prepare_job *job, const char **filenames, int n,
                                                    int ok, int with_cached, int *got);
// every track of the cache except the n named files, sorted by name; consumes the job
FilenameHashprintPair *hpfw_internal_prepare_finish_cached(hpfw_legacy_collector *c, hpfw_prepare_job *job, const char **filenames, int n,
                                                           int *got);
}
