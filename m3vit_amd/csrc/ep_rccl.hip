// Expert-parallel exchange over RCCL behind the C ABI (m3_ep_init / m3_ep_exchange_counts / m3_ep_dispatch /
// m3_ep_return / m3_ep_destroy): what fastmoe's expert_exchange / global_scatter / global_gather do behind
// _fmoe_general_global_forward (reference call site models/moe/ckpt/custom_moe_layer.py:263-265, world_size > 1; experts
// sharded per utils/common_config.py:179-185), for a caller that does not go through torch.distributed.
//   - xGMI is a point-to-point mesh: the all-to-all-v is ONE grouped set of ncclSend / ncclRecv pairs, a distinct peer per
//     link, on the caller's stream (ncclGroupStart .. ncclGroupEnd); nothing is chunked into ring steps;
//   - the library keeps no state but the communicators (a small table of handles); buffers and split sizes are the caller's;
//   - librccl is opened at the first m3_ep_unique_id / m3_ep_init call (dlopen), so libm3vit_hip.so itself has no link-time
//     dependency on it and single-GPU users never load it.
// The engine's default exchange stays torch.distributed (backend nccl = RCCL), which the two-rank gloo rehearsals can test;
// these entry points are exercised here with a one-rank communicator (tests/test_ep_rccl_gpu.py) - this build box has one GPU.
#include <dlfcn.h>

#include "common.h"

namespace m3 {

typedef struct { char internal[128]; } rccl_unique_id;         // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm;
typedef int rccl_result;                                        // ncclResult_t: 0 = ncclSuccess
enum { RCCL_INT8 = 0, RCCL_INT64 = 4 };                         // ncclDataType_t: ncclInt8 = 0, ncclInt64 = 4

struct RcclApi {
  void *so = nullptr;
  rccl_result (*GetUniqueId)(rccl_unique_id *) = nullptr;
  rccl_result (*CommInitRank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
  rccl_result (*CommDestroy)(rccl_comm) = nullptr;
  rccl_result (*GroupStart)() = nullptr;
  rccl_result (*GroupEnd)() = nullptr;
  rccl_result (*Send)(const void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
  rccl_result (*Recv)(void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(rccl_result) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
  if (g_rccl.so) return M3_OK;
  void *so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!so) { set_error("m3_ep: cannot open librccl (%s)", dlerror()); return M3_ERR_UNSUPPORTED; }
#define M3_SYM(field, name)                                                          \
  *(void **)(&g_rccl.field) = dlsym(so, name);                                       \
  if (!g_rccl.field) { set_error("m3_ep: librccl has no %s", name); dlclose(so); return M3_ERR_UNSUPPORTED; }
  M3_SYM(GetUniqueId, "ncclGetUniqueId")
  M3_SYM(CommInitRank, "ncclCommInitRank")
  M3_SYM(CommDestroy, "ncclCommDestroy")
  M3_SYM(GroupStart, "ncclGroupStart")
  M3_SYM(GroupEnd, "ncclGroupEnd")
  M3_SYM(Send, "ncclSend")
  M3_SYM(Recv, "ncclRecv")
  M3_SYM(GetErrorString, "ncclGetErrorString")
#undef M3_SYM
  g_rccl.so = so;
  return M3_OK;
}

constexpr int EP_MAX_COMMS = 16;
struct EpComm { rccl_comm comm; int rank, world; bool live; };
static EpComm g_comms[EP_MAX_COMMS];

static int rccl_check(rccl_result r, const char *what) {
  if (r == 0) return M3_OK;
  set_error("%s: RCCL error %d (%s)", what, r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return M3_ERR_LAUNCH;
}

static EpComm *ep_get(int handle, const char *what) {
  if (handle < 0 || handle >= EP_MAX_COMMS || !g_comms[handle].live) {
    set_error("%s: bad communicator handle %d", what, handle);
    return nullptr;
  }
  return &g_comms[handle];
}

// all-to-all-v of bytes: peer p gets send[in_off[p] .. + in_n[p]) and fills recv[out_off[p] .. + out_n[p]) (counts in bytes)
static int ep_a2av(EpComm *c, const char *send, const int64_t *in_n, char *recv, const int64_t *out_n, int64_t unit, hipStream_t s,
                   const char *what) {
  int rc = rccl_check(g_rccl.GroupStart(), what);
  if (rc) return rc;
  int64_t io = 0, oo = 0;
  for (int p = 0; p < c->world; ++p) {
    if (in_n[p] > 0) { rc = rccl_check(g_rccl.Send(send + io * unit, (size_t)(in_n[p] * unit), RCCL_INT8, p, c->comm, s), what); if (rc) break; }
    if (out_n[p] > 0) { rc = rccl_check(g_rccl.Recv(recv + oo * unit, (size_t)(out_n[p] * unit), RCCL_INT8, p, c->comm, s), what); if (rc) break; }
    io += in_n[p]; oo += out_n[p];
  }
  const int rc2 = rccl_check(g_rccl.GroupEnd(), what);
  return rc ? rc : rc2;
}

}  // namespace m3

using namespace m3;

extern "C" int m3_ep_unique_id(void *out128) {
  M3_REQUIRE(out128, "m3_ep_unique_id: null output");
  int rc = rccl_load();
  if (rc) return rc;
  rccl_unique_id id;
  rc = rccl_check(g_rccl.GetUniqueId(&id), "m3_ep_unique_id");
  if (rc) return rc;
  memcpy(out128, id.internal, 128);
  return M3_OK;
}

extern "C" int m3_ep_init(const void *unique_id128, int rank, int world, int *handle) {
  M3_REQUIRE(unique_id128 && handle, "m3_ep_init: null argument");
  M3_REQUIRE(world >= 1 && rank >= 0 && rank < world, "m3_ep_init: rank %d of %d", rank, world);
  int rc = rccl_load();
  if (rc) return rc;
  int h = -1;
  for (int i = 0; i < EP_MAX_COMMS; ++i)
    if (!g_comms[i].live) { h = i; break; }
  M3_REQUIRE(h >= 0, "m3_ep_init: more than %d live communicators", EP_MAX_COMMS);
  rccl_unique_id id;
  memcpy(id.internal, unique_id128, 128);
  rccl_comm comm = nullptr;
  rc = rccl_check(g_rccl.CommInitRank(&comm, world, id, rank), "m3_ep_init");
  if (rc) return rc;
  g_comms[h].comm = comm; g_comms[h].rank = rank; g_comms[h].world = world; g_comms[h].live = true;
  *handle = h;
  return M3_OK;
}

extern "C" int m3_ep_destroy(int handle) {
  EpComm *c = ep_get(handle, "m3_ep_destroy");
  if (!c) return M3_ERR_ARG;
  const int rc = rccl_check(g_rccl.CommDestroy(c->comm), "m3_ep_destroy");
  c->live = false; c->comm = nullptr;
  return rc;
}

// send_counts / recv_counts: device int64 [world * e_loc]; entry d * e_loc + e of send = rows this rank routes to local
// expert e of rank d; entry s * e_loc + e of recv = rows rank s routes to this rank's local expert e (fastmoe expert_exchange)
extern "C" int m3_ep_exchange_counts(int handle, const int64_t *send_counts, int64_t *recv_counts, int e_loc, void *stream) {
  EpComm *c = ep_get(handle, "m3_ep_exchange_counts");
  if (!c) return M3_ERR_ARG;
  M3_REQUIRE(send_counts && recv_counts && e_loc >= 1, "m3_ep_exchange_counts: bad arguments");
  int64_t n[64];
  M3_REQUIRE(c->world <= 64, "m3_ep_exchange_counts: world > 64");
  for (int p = 0; p < c->world; ++p) n[p] = e_loc;
  return ep_a2av(c, (const char *)send_counts, n, (char *)recv_counts, n, 8, (hipStream_t)stream, "m3_ep_exchange_counts");
}

// rows: [sum(in_splits), row_bytes] expert-major by destination -> [sum(out_splits), row_bytes] by source (global_scatter);
// in_splits / out_splits are HOST arrays of `world` row counts (what m3_ep_plan hands back through its one host read)
extern "C" int m3_ep_dispatch(int handle, const void *send_rows, const int64_t *in_splits, void *recv_rows, const int64_t *out_splits,
                              int64_t row_bytes, void *stream) {
  EpComm *c = ep_get(handle, "m3_ep_dispatch");
  if (!c) return M3_ERR_ARG;
  M3_REQUIRE(in_splits && out_splits && row_bytes > 0, "m3_ep_dispatch: bad arguments");
  int64_t ti = 0, to = 0;
  for (int p = 0; p < c->world; ++p) {
    M3_REQUIRE(in_splits[p] >= 0 && out_splits[p] >= 0, "m3_ep_dispatch: negative split");
    ti += in_splits[p]; to += out_splits[p];
  }
  M3_REQUIRE((ti == 0 || send_rows) && (to == 0 || recv_rows), "m3_ep_dispatch: null buffer");
  return ep_a2av(c, (const char *)send_rows, in_splits, (char *)recv_rows, out_splits, row_bytes, (hipStream_t)stream, "m3_ep_dispatch");
}

// the way home (global_gather): the same exchange with the roles of the split vectors swapped
extern "C" int m3_ep_return(int handle, const void *send_rows, const int64_t *out_splits, void *recv_rows, const int64_t *in_splits,
                            int64_t row_bytes, void *stream) {
  return m3_ep_dispatch(handle, send_rows, out_splits, recv_rows, in_splits, row_bytes, stream);
}
