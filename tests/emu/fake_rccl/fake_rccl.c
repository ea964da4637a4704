/* TEST INFRASTRUCTURE ONLY — a stand-in for the handful of RCCL entry points the plugin's sharded group-by resolves
 * with dlsym (arrow_amd/csrc/plugin/sharded.inc), so that its world_size-2 exchange runs in the GPU-less CPU tier:
 * ranks are processes, "device" memory is host memory (tests/emu), a message is a file in a directory named by the
 * unique id (written under a temporary name, then renamed: the receiver never sees half a message).  Sends never
 * block; inside a group every send goes out before the first receive is waited for. */
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* -DFAKE_RCCL_DEVICE (the GPU tier's world-2-on-ONE-GPU test, tests/test_sharded_rccl_plugin.py): the buffers are HBM.
 * A send waits for the caller's stream, stages the bytes through host memory and writes the file; a receive reads the
 * file and copies it up with a blocking hipMemcpy (complete before anything enqueued afterwards runs).  The real librccl
 * refuses two ranks on one device; this lets the C++ exchange run with a peer that is not itself on real kernels. */
#ifdef FAKE_RCCL_DEVICE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#endif

typedef struct { char bytes[128]; } ncclUniqueId;
typedef struct FakeComm {
  int nranks, rank;
  char dir[128];
  uint64_t send_seq[64], recv_seq[64];
} FakeComm;
typedef struct Op { int is_send; void* buf; size_t bytes; int peer; FakeComm* comm; } Op;
static __thread Op g_ops[256];
static __thread int g_nops = 0, g_depth = 0;

static size_t type_size(int dtype) {
  switch (dtype) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; default: return 1; }
}
static void msg_path(FakeComm* c, int src, int dst, uint64_t seq, const char* suffix, char* out, size_t n) {
  snprintf(out, n, "%s/m_%d_%d_%llu%s", c->dir, src, dst, (unsigned long long)seq, suffix);
}
static int do_send_host(FakeComm* c, const void* buf, size_t bytes, int peer);
static int do_recv_host(FakeComm* c, void* buf, size_t bytes, int peer);
#ifdef FAKE_RCCL_DEVICE
static int do_send(FakeComm* c, const void* buf, size_t bytes, int peer) {
  void* host = malloc(bytes ? bytes : 1);
  if (!host) return 2;
  if (bytes && hipMemcpy(host, buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(host); return 2; }
  const int rc = do_send_host(c, host, bytes, peer);
  free(host);
  return rc;
}
static int do_recv(FakeComm* c, void* buf, size_t bytes, int peer) {
  void* host = malloc(bytes ? bytes : 1);
  if (!host) return 2;
  int rc = do_recv_host(c, host, bytes, peer);
  if (!rc && bytes && hipMemcpy(buf, host, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 2;
  free(host);
  return rc;
}
#else
static int do_send(FakeComm* c, const void* buf, size_t bytes, int peer) { return do_send_host(c, buf, bytes, peer); }
static int do_recv(FakeComm* c, void* buf, size_t bytes, int peer) { return do_recv_host(c, buf, bytes, peer); }
#endif
static int do_send_host(FakeComm* c, const void* buf, size_t bytes, int peer) {
  char tmp[256], fin[256];
  const uint64_t seq = c->send_seq[peer]++;
  msg_path(c, c->rank, peer, seq, ".tmp", tmp, sizeof tmp);
  msg_path(c, c->rank, peer, seq, "", fin, sizeof fin);
  FILE* f = fopen(tmp, "wb");
  if (!f) return 2;
  if (bytes && fwrite(buf, 1, bytes, f) != bytes) { fclose(f); return 2; }
  fclose(f);
  return rename(tmp, fin) == 0 ? 0 : 2;
}
static int do_recv_host(FakeComm* c, void* buf, size_t bytes, int peer) {
  char fin[256];
  const uint64_t seq = c->recv_seq[peer]++;
  msg_path(c, peer, c->rank, seq, "", fin, sizeof fin);
  for (int spins = 0; spins < 600000; ++spins) {      /* <= 10 minutes */
    FILE* f = fopen(fin, "rb");
    if (f) {
      const size_t got = bytes ? fread(buf, 1, bytes, f) : 0;
      fclose(f);
      unlink(fin);
      return got == bytes ? 0 : 2;
    }
    struct timespec ts = {0, 1000000};
    nanosleep(&ts, NULL);
  }
  return 2;
}
static int flush_ops(void) {
  int rc = 0;
  for (int i = 0; i < g_nops && !rc; ++i) if (g_ops[i].is_send) rc = do_send(g_ops[i].comm, g_ops[i].buf, g_ops[i].bytes, g_ops[i].peer);
  for (int i = 0; i < g_nops && !rc; ++i) if (!g_ops[i].is_send) rc = do_recv(g_ops[i].comm, g_ops[i].buf, g_ops[i].bytes, g_ops[i].peer);
  g_nops = 0;
  return rc;
}

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->bytes, sizeof id->bytes, "/tmp/arx_fake_rccl_%d_%ld", (int)getpid(), (long)time(NULL));
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > 64) return 4;
  FakeComm* c = (FakeComm*)calloc(1, sizeof *c);
  c->nranks = nranks;
  c->rank = rank;
  snprintf(c->dir, sizeof c->dir, "%s", id.bytes);
  if (mkdir(c->dir, 0700) != 0 && errno != EEXIST) { free(c); return 2; }
  *comm = c;
  return 0;
}
int ncclCommDestroy(void* comm) {
  FakeComm* c = (FakeComm*)comm;
  if (c && c->rank == 0) rmdir(c->dir);   /* (fails harmlessly while a peer's message is still there) */
  free(c);
  return 0;
}
int ncclGroupStart(void) { ++g_depth; return 0; }
static void wait_stream(void* stream) {
#ifdef FAKE_RCCL_DEVICE
  (void)hipStreamSynchronize((hipStream_t)stream);   /* what the caller enqueued before the call is what gets sent */
#else
  (void)stream;
#endif
}
int ncclGroupEnd(void) { return --g_depth == 0 ? flush_ops() : 0; }
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
  wait_stream(stream);
  if (g_depth > 0) { g_ops[g_nops++] = (Op){1, (void*)buf, count * type_size(dtype), peer, (FakeComm*)comm}; return 0; }
  return do_send((FakeComm*)comm, buf, count * type_size(dtype), peer);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
  wait_stream(stream);
  if (g_depth > 0) { g_ops[g_nops++] = (Op){0, buf, count * type_size(dtype), peer, (FakeComm*)comm}; return 0; }
  return do_recv((FakeComm*)comm, buf, count * type_size(dtype), peer);
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
  wait_stream(stream);
  FakeComm* c = (FakeComm*)comm;
  const size_t bytes = count * type_size(dtype);
  int rc = 0;
  for (int p = 0; p < c->nranks && !rc; ++p) rc = do_send(c, send, bytes, p);
  for (int p = 0; p < c->nranks && !rc; ++p) rc = do_recv(c, (char*)recv + (size_t)p * bytes, bytes, p);
  return rc;
}
const char* ncclGetErrorString(int rc) { return rc == 2 ? "fake RCCL: message file error / peer timeout" : "fake RCCL: bad argument"; }
