// r3d_rccl.h -- the engine's one door to RCCL (host code; included by r3d_engine.hip only).
//
// librccl is bound at first use, not at link time: a process that already has an RCCL in it -- a Python
// host in which torch.distributed loaded torch's own copy -- must go on using THAT library for every
// communicator (two copies of the library in one process each keep their own bootstrap state and
// topology caches, and which one a link-time dependency binds to depends on import order).  So:
// the library already loaded under the soname librccl.so.1 if there is one, else the loader's search
// path, else /opt/rocm/lib.  Which one was bound and its version are reported (r3d_comm_describe).
//
// reduce_block() below is the ONLY place the engine issues a data-path collective: the three buffers of
// one rank's result block (f64 energies, u64 counts, u64 counters) summed over the communicator's ranks
// -- to one root (ncclReduce: a node's shards to shard 0, r3d_node_run) or to every rank (ncclAllReduce:
// one process per GPU, r3d_comm_reduce).  This is the reference's "replicas + combine"
// (scripts/do-parallel.sh:23-29, vis/seisplot/combine.m:26-33: the replicas' traces add) with the combine
// done on the devices.
#ifndef R3D_RCCL_H_
#define R3D_RCCL_H_

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and constants only: no symbol of the library is referenced

#include <mutex>
#include <string>

namespace r3d {

struct Rccl {
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  std::string where;     // how the library was found
  std::string problem;   // why it was not
  int version = 0;
};

inline const Rccl* rccl(std::string* why_not = nullptr) {
  static Rccl R;
  static bool ok = false;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    R.where = "librccl.so.1 (already in the process)";
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL), R.where = "librccl.so.1 (loader search path)";
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL), R.where = "/opt/rocm/lib/librccl.so.1";
    if (!h) {
      const char* e = dlerror();
      R.problem = std::string("librccl.so.1 cannot be loaded") + (e ? std::string(": ") + e : std::string());
      return;
    }
    bool all = true;
    auto sym = [&](auto& fn, const char* name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
      if (!fn) all = false, R.problem = std::string("librccl has no ") + name;
    };
    sym(R.GetVersion, "ncclGetVersion"), sym(R.GetUniqueId, "ncclGetUniqueId"), sym(R.CommInitRank, "ncclCommInitRank");
    sym(R.CommInitAll, "ncclCommInitAll"), sym(R.CommDestroy, "ncclCommDestroy"), sym(R.CommAbort, "ncclCommAbort");
    sym(R.CommCount, "ncclCommCount"), sym(R.CommUserRank, "ncclCommUserRank"), sym(R.GetErrorString, "ncclGetErrorString");
    sym(R.GroupStart, "ncclGroupStart"), sym(R.GroupEnd, "ncclGroupEnd"), sym(R.Reduce, "ncclReduce");
    sym(R.AllReduce, "ncclAllReduce");
    if (!all) return;
    {
      Dl_info info;   // (the file the symbols came from, where the loader can say)
      if (dladdr(reinterpret_cast<void*>(R.AllReduce), &info) && info.dli_fname) R.where += std::string(" = ") + info.dli_fname;
    }
    (void)R.GetVersion(&R.version);
    ok = true;
  });
  if (!ok && why_not) *why_not = R.problem;
  return ok ? &R : nullptr;
}

// One rank's result block summed over the ranks of `comm`, in stream order on `s`; root >= 0: the sums land in the
// root's *_out (the other ranks' *_out is not written: they may name their own block), root < 0: in every rank's.
// Call between GroupStart / GroupEnd when one thread speaks for several ranks.
inline ncclResult_t reduce_block(const Rccl& R, ncclComm_t comm, hipStream_t s, int root, const void* e_in, void* e_out, size_t ne,
                                 const void* c_in, void* c_out, size_t nc, const void* s_in, void* s_out, size_t ns) {
  auto one = [&](const void* in, void* out, size_t n, ncclDataType_t t) {
    if (!n) return ncclSuccess;
    return root >= 0 ? R.Reduce(in, out, n, t, ncclSum, root, comm, s) : R.AllReduce(in, out, n, t, ncclSum, comm, s);
  };
  ncclResult_t r = one(e_in, e_out, ne, ncclDouble);
  if (r == ncclSuccess) r = one(c_in, c_out, nc, ncclUint64);
  if (r == ncclSuccess) r = one(s_in, s_out, ns, ncclUint64);
  return r;
}

}  // namespace r3d
#endif
