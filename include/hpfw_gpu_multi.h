/*
 * hpfw_gpu_multi.h -- C-ABI of the multi-GPU host path (libhpfw_gpu_multi.so = this API on top of
 * libhpfw_gpu.so and librccl.so).  One process drives the GPUs of one node: one hpfw_gpu handle per
 * device, one host thread per device while work is being enqueued.
 *
 * What it replaces (paths relative to the hpfw reference tree):
 *   MemoryStorage::build / find          include/hpfw/audioproblems/live-song-id/storage.h:21-64
 *   as LiveSongIdentification calls them include/hpfw/audioproblems/live-song-id/live_song_id.h:31-54
 *   ParallelCollector::collect_fingerprints' parallel_for over files
 *                                        include/hpfw/core/parallel_collector.h:115-137
 *   preprocess' `accum_cov += cov` under a mutex (:93-97) -> one ncclAllReduce over the devices
 *
 * Partitioning: the index is sharded per audio file -- clip i of n lives on the
 * shard whose contiguous block [lo, hi) holds it (block sizes differ by at most one, earlier shards take the
 * extra) and is reported under its global id (hpfw_gpu_index_set_clip_base).  Queries are replicated.
 * Every shard scans its block and keeps its own top-k; the single exchange step is ONE ncclAllGather of
 * n_q x k x 16 bytes per shard over xGMI, then the same deterministic merge by (dist, clip) as
 * hpfw_gpu_merge_topk -- the result does not depend on the number of shards.  Extraction shards clips the
 * same way and needs no collective.
 *
 * When several shards are placed on ONE device (devices[] repeats an ordinal: tests on a one-GPU box, or an
 * index split for capacity), RCCL cannot give each its own rank (it refuses two ranks on a device): the
 * communicators span the distinct devices, the shards of a device write their lists side by side into that
 * device's send buffer, and the same single ncclAllGather moves them (every device must then hold the same
 * number of shards).
 *
 * Conventions as in hpfw_gpu.h: every function returns 0 or a negative hpfw_status, never throws; messages
 * through hpfw_gpu_last_error().  All pointers are HOST pointers.
 */
#ifndef HPFW_GPU_MULTI_H
#define HPFW_GPU_MULTI_H

#include "hpfw_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpfw_gpu_group hpfw_gpu_group; /* opaque */

/* n_shards handles, shard i on device devices[i]; devices == NULL: shard i on device i.
 * Communicators: ncclCommInitAll over the distinct devices (also for one device: world size 1). */
int hpfw_gpu_group_create(const int *devices, int n_shards, hpfw_gpu_group **out);
/* placement from the environment: HPFW_GPU_DEVICES = comma-separated device ordinals, one per shard
 * ("0,1,2,3,4,5,6,7"; an ordinal may repeat); unset = one shard on every visible device */
int hpfw_gpu_group_create_env(hpfw_gpu_group **out);
void hpfw_gpu_group_destroy(hpfw_gpu_group *g);
int hpfw_gpu_group_size(const hpfw_gpu_group *g);
/* the single-device handle of shard i (owned by the group) */
hpfw_gpu *hpfw_gpu_group_handle(hpfw_gpu_group *g, int shard);
/* "rccl" when the exchange step runs ncclAllGather, "rccl+local" when some shards share a device */
const char *hpfw_gpu_group_exchange(const hpfw_gpu_group *g);

/* filters replicated to every shard (620 KB); layout as hpfw_gpu_set_filters */
int hpfw_gpu_group_set_filters(hpfw_gpu_group *g, const float *filters_colmajor);

/* calc_hashprint for n_clips clips of n_samples samples: clips sharded contiguously over the shards, one
 * host thread per shard, no collective.  hp [n_clips][n_hp] in input order. */
int hpfw_gpu_group_extract_pcm16(hpfw_gpu_group *g, const int16_t *pcm, int64_t n_samples, int64_t n_clips,
                                 uint64_t *hp);

/* MemoryStorage::build (storage.h:21-25) sharded: replaces the index.  Clip i is hp[offsets[i] ..
 * offsets[i+1]); shard s receives its contiguous block with clip base = the block's first global id. */
int hpfw_gpu_group_index_build(hpfw_gpu_group *g, const uint64_t *hp, const int64_t *offsets, int64_t n_clips);
int64_t hpfw_gpu_group_index_size(const hpfw_gpu_group *g);
/* [lo, hi) of shard s for an index of n_clips clips (host-only arithmetic) */
void hpfw_gpu_shard_range(int64_t n_clips, int shard, int n_shards, int64_t *lo, int64_t *hi);

/* MemoryStorage::find / the notebook's top-k (storage.h:27-64, liveid.ipynb cell 9) over the sharded index:
 * replicated queries -> per-shard scan + top-k -> ncclAllGather of the per-shard lists -> merge.
 * out [n_q][k], ascending (dist, global clip id); identical to hpfw_gpu_search_topk on the unsharded index. */
int hpfw_gpu_group_search_topk(hpfw_gpu_group *g, const uint64_t *q_hp, const int64_t *q_off, int64_t n_q, int k,
                               hpfw_hit *out);

/* filter learning over the shards (parallel_collector.h:82-112): every shard accumulates the covariance of
 * its block of clips; learn = ncclAllReduce(sum) of the 2420 x 2420 matrices (23.4 MB) over the devices, the
 * eigen-solve on the host, the filters installed on every shard.  filters_colmajor_out may be NULL. */
int hpfw_gpu_group_cov_reset(hpfw_gpu_group *g);
int hpfw_gpu_group_cov_accumulate_pcm16(hpfw_gpu_group *g, const int16_t *pcm, int64_t n_samples, int64_t n_clips);
int hpfw_gpu_group_learn_filters(hpfw_gpu_group *g, float *filters_colmajor_out);

/* ---- ParallelCollector over the shards (parallel_collector.h:48-73): index() end to end on the node's GPUs ----
 * prepare: the files are sharded contiguously; every shard (its own host thread, collector and device) reads its
 * files, computes their spectrograms (cached under <cache>/spectros/ as cache.h:30-33 does) and adds their frame
 * covariances to its accum_cov; ONE ncclAllReduce sums accum_cov over the devices; shard 0 solves for the filters
 * and writes filters.cereal / accum_cov.cereal; every shard then hashes the spectrograms it kept on its device.
 * Results: this call's files in input order (failed files dropped, :101-103), then every other track of the
 * cache, sorted (collect_fingerprints walks the whole cache, :115-137); release with prepare_result_free.
 * HPFW_PREPARE_KEEP_FILTERS=1 keeps loaded filters instead of learning.  cache: NULL / "" = "cache/". */
int hpfw_gpu_group_load(hpfw_gpu_group *g, const char *cache);  /* ParallelCollector::load on every shard */
int hpfw_gpu_group_save(hpfw_gpu_group *g, const char *cache);  /* ParallelCollector::save (shard 0 holds accum_cov) */
FilenameHashprintPair *hpfw_gpu_group_prepare(hpfw_gpu_group *g, const char **filenames, int n, int *got);
/* calc_hashprint (parallel_collector.h:54-59) on shard 0; release with calc_hashprint_result_free */
uint64_t *hpfw_gpu_group_calc_hashprint(hpfw_gpu_group *g, const char *filename, int *size);

#ifdef __cplusplus
}
#endif
#endif
