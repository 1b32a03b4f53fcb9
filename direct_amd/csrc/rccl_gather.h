// Config-5 reduction across the GPUs of a node for a C / C++ host (BASELINE.json configs[4]):
// the cheapest feasible trajectory of the whole sharded batch, materialised on every rank.
//
// The reference has no counterpart (single process, no collectives).  The solve itself shards with no
// data-path exchange; this is the one real exchange step, and it is tiny: one ncclAllGather of a 16-byte
// (cost, global index) record per rank and one of each rank's local-best block (n_seg_max x 19 reals,
// 7.6 kB in float at N = 100).  Everything runs on the handle's stream; xGMI is latency-bound at these sizes, so a
// single all-gather of the blocks (no data-dependent root, no second round trip) is the right shape.
//
// RCCL is resolved at run time (dlsym on the process first, then dlopen of librccl.so.1): the library does
// not link against it, so single-GPU users need no RCCL, and inside a process that already carries an RCCL
// (PyTorch bundles one under the same soname) the SAME copy is used.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdint.h>

namespace direct {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

inline const RcclApi& rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    void* lib = nullptr;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(RTLD_DEFAULT, name);
      if (!p) {
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) p = dlsym(lib, name);
      }
      return p;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.GetErrorString;
    return a;
  }();
  return api;
}

struct BestRec {  // one per rank in the gathered array
  double cost;
  long long index;  // global problem index, -1 when the rank has no feasible trajectory
};

// slot `rank` of the two gather buffers <- this rank's local best (idx / cost from k_best, device memory)
template <typename Real>
__global__ void k_pack_best(const int* best_idx, const double* best_cost, long long first_index, int rank, int nmax,
                            const Real* bez, const Real* T, BestRec* recs, Real* blocks) {
  const int li = *best_idx;
  Real* blk = blocks + (size_t)rank * nmax * 19;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    recs[rank].cost = li >= 0 ? *best_cost : INFINITY;
    recs[rank].index = li >= 0 ? first_index + li : -1;
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nmax * 19; e += gridDim.x * blockDim.x) {
    Real v = (Real)0;
    if (li >= 0) v = e < nmax * 18 ? bez[(size_t)li * nmax * 18 + e] : T[(size_t)li * nmax + (e - nmax * 18)];
    blk[e] = v;
  }
}

// argmin over the gathered records (ties: the smaller global index, so every rank picks the same winner
// whatever the rank order) and the winner's block -> the caller's arrays
template <typename Real>
__global__ void k_pick_best(const BestRec* recs, const Real* blocks, int n_ranks, int nmax, BestRec* winner, int* owner,
                            Real* out_bez, Real* out_T) {
  int w = -1;
  for (int r = 0; r < n_ranks; r++) {
    if (recs[r].index < 0) continue;
    if (w < 0 || recs[r].cost < recs[w].cost || (recs[r].cost == recs[w].cost && recs[r].index < recs[w].index)) w = r;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    winner->cost = w >= 0 ? recs[w].cost : INFINITY;
    winner->index = w >= 0 ? recs[w].index : -1;
    *owner = w;
  }
  const Real* blk = blocks + (size_t)(w >= 0 ? w : 0) * nmax * 19;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nmax * 19; e += gridDim.x * blockDim.x) {
    const Real v = w >= 0 ? blk[e] : (Real)0;
    if (e < nmax * 18) {
      if (out_bez) out_bez[e] = v;
    } else if (out_T) {
      out_T[e - nmax * 18] = v;
    }
  }
}

}  // namespace direct
