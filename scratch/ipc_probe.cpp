// ipc_probe.cpp -- round-5 feasibility probe for the peer-memory transport (NOT product code).
// N processes (forked before any HIP call) share device 0, export one arena each through hipIpc, and run
//   (1) a ring push: payload into the upper neighbour's arena + sequence flag, bounded poll on the own flag, every word checked
//   (2) a mailbox all-reduce in one single-workgroup kernel
//   (3) hipStreamWriteValue64 / hipStreamWaitValue64 on the same memory (return codes, then timing if accepted)
// usage: ipc_probe NRANKS ITERS BYTES MEMKIND(0 hipMalloc, 1 fine-grained, 2 uncached)
// IPC_PROBE_DISTINCT=1: rank r binds device r mod (device count) instead of device 0 -- the first-contact form for a multi-GPU node
// (scratch/first_contact.sh): the same pushes, flags and mailboxes, every word checked, now across xGMI.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <thread>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "[%d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #e, hipGetErrorString(r_)); fflush(stderr); _exit(3); } } while (0)
static int g_rank = -1;
enum { MAXR = 8, CTRL_WORDS = 512 };

struct Shm {
  volatile long gen[MAXR];
  hipIpcMemHandle_t h[MAXR];
};

static int probe_device(int rank) {
  const char *e = getenv("IPC_PROBE_DISTINCT");
  if (!e || atoi(e) == 0) return 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
  return rank % n;
}
static long g_gen = 0;
static void hbarrier(Shm *s, int n, int rank, int id) {
  const long my = ++g_gen;
  __atomic_store_n(&s->gen[rank], my, __ATOMIC_RELEASE);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < n; r++)
    while (__atomic_load_n(&s->gen[r], __ATOMIC_ACQUIRE) < my) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { fprintf(stderr, "[%d] host barrier %d timed out\n", rank, id); _exit(4); }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// control words (uint64) at the head of each arena: [0] data flag, [1] credit, [2] error, [8..8+MAXR) mailbox flags, payload from byte 4096
__device__ inline bool poll_ge(const unsigned long long *p, unsigned long long want, unsigned long long *err) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 300000000LL) { __hip_atomic_store(err, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }   // 3 s at 100 MHz
  }
  return true;
}

__global__ void k_push(const double2 *src, double2 *dst, size_t n2, unsigned long long *peer_flag, unsigned long long seq, unsigned int *done,
                       const unsigned long long *credit, unsigned long long *err) {
  __shared__ int okp;
  if (threadIdx.x == 0) okp = poll_ge(credit, seq - 1, err) ? 1 : 0;      // the receiver has unpacked exchange seq-1: its arena is free
  __syncthreads();
  if (!okp) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int k = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (k == gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(peer_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void k_fill(double2 *p, size_t n2, double tag) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(tag + (double)i, -tag);
}

// wait for the flag (one lane per block polls), then check every word of the payload, then copy it out (the unpack)
__global__ void k_wait_check(const double2 *arena, double2 *out, size_t n2, unsigned long long *flag, unsigned long long seq, double tag,
                             unsigned long long *err, unsigned long long *bad, unsigned int *done, unsigned long long *sender_credit) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = poll_ge(flag, seq, err) ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (!ok) return;
  unsigned long long nb = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    double2 v = arena[i];
    if (v.x != tag + (double)i || v.y != -tag) nb++;
    out[i] = v;
  }
  if (nb) atomicAdd(bad, nb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int k = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (k == gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sender_credit, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// mailbox all-reduce, one workgroup: x[0..n) -> every rank's mailbox [slot][me][n]; wait N flags; sum in rank order
__global__ void k_allreduce(double *x, int n, int nranks, int me, double *const *mbox /*[nranks] peer-mapped payload bases*/,
                            unsigned long long *const *mflag /*[nranks] peer-mapped flag arrays*/, double *my_mbox, unsigned long long *my_flag,
                            unsigned long long seq, int stride, unsigned long long *err) {
  const int slot = (int)(seq & 3);
  for (int r = 0; r < nranks; r++)
    for (int i = threadIdx.x; i < n; i += blockDim.x) mbox[r][((size_t)slot * MAXR + me) * stride + i] = x[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < nranks) __hip_atomic_store(&mflag[threadIdx.x][slot * MAXR + me], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < nranks) {
    if (!poll_ge(&my_flag[slot * MAXR + threadIdx.x], seq, err)) ok = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (!ok) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double s = 0;
    for (int r = 0; r < nranks; r++) s += my_mbox[((size_t)slot * MAXR + r) * stride + i];
    x[i] = s;
  }
}

// regrow mode: what peer_ensure_arena does, 24 times with growing sizes: close peers' mappings, free (mode 1) or keep (mode 2) the old
// arena, allocate a bigger one, export, open the neighbours'
static int regrow(int rank, int n, int mode, Shm *shm) {
  g_rank = rank;
  CK(hipSetDevice(probe_device(rank)));
  char *arena = nullptr, *peer[MAXR] = {nullptr};
  size_t sz = (size_t)1 << 20;
  int rc = 0;
  for (int it = 0; it < 24; it++, sz = sz + sz / 2 + 4096 * (size_t)it) {
    hbarrier(shm, n, rank, 100);
    for (int r = 0; r < n; r++) if (peer[r]) { hipError_t e = hipIpcCloseMemHandle(peer[r]); if (e != hipSuccess) printf("[%d] it %d close(%d): %s\n", rank, it, r, hipGetErrorString(e)); peer[r] = nullptr; }
    hbarrier(shm, n, rank, 101);
    if (arena && mode == 1) CK(hipFree(arena));
    arena = nullptr;
    CK(hipMalloc((void **)&arena, 2 * sz));
    hipError_t e = hipIpcGetMemHandle(&shm->h[rank], arena);
    if (e != hipSuccess) { printf("[%d] it %d size %zu ptr %p: hipIpcGetMemHandle: %s\n", rank, it, 2 * sz, (void *)arena, hipGetErrorString(e)); fflush(stdout); rc = 6; (void)hipGetLastError(); }
    hbarrier(shm, n, rank, 102);
    if (rc) return rc;
    for (int k = -1; k <= 1; k += 2) {
      const int r = (rank + k + n) % n;
      if (r == rank || peer[r]) continue;
      e = hipIpcOpenMemHandle((void **)&peer[r], shm->h[r], hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { printf("[%d] it %d open(%d): %s\n", rank, it, r, hipGetErrorString(e)); fflush(stdout); return 7; }
    }
    CK(hipMemset(peer[(rank + 1) % n], rank + 1, 2 * sz));
    CK(hipDeviceSynchronize());
    hbarrier(shm, n, rank, 103);
  }
  printf("[%d] regrow mode %d: 24 growths up to %zu bytes ok\n", rank, mode, 2 * sz);
  return 0;
}

static int child(int rank, int n, int iters, size_t bytes, int memkind, Shm *shm) {
  g_rank = rank;
  CK(hipSetDevice(probe_device(rank)));
  const size_t mbox_bytes = 4 * MAXR * 2048 * sizeof(double);
  const size_t arena_bytes = 4096 + mbox_bytes + bytes;
  char *arena = nullptr;
  hipError_t e;
  if (memkind == 0) e = hipMalloc((void **)&arena, arena_bytes);
  else if (memkind == 1) e = hipExtMallocWithFlags((void **)&arena, arena_bytes, hipDeviceMallocFinegrained);
  else e = hipExtMallocWithFlags((void **)&arena, arena_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { printf("[%d] alloc kind %d failed: %s\n", rank, memkind, hipGetErrorString(e)); return 5; }
  CK(hipMemset(arena, 0, arena_bytes));
  CK(hipDeviceSynchronize());
  e = hipIpcGetMemHandle(&shm->h[rank], arena);
  if (e != hipSuccess) { printf("[%d] hipIpcGetMemHandle kind %d: %s\n", rank, memkind, hipGetErrorString(e)); return 6; }
  hbarrier(shm, n, rank, 0);
  char *peer[MAXR];
  for (int r = 0; r < n; r++) {
    if (r == rank) { peer[r] = arena; continue; }
    e = hipIpcOpenMemHandle((void **)&peer[r], shm->h[r], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { printf("[%d] hipIpcOpenMemHandle(%d) kind %d: %s\n", rank, r, memkind, hipGetErrorString(e)); return 7; }
  }
  hbarrier(shm, n, rank, 1);
  if (rank == 0) printf("ipc open ok: %d ranks, kind %d, arena %zu bytes\n", n, memkind, arena_bytes);

  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t n2 = bytes / 16;
  double2 *src, *out;
  unsigned int *done;
  unsigned long long *bad;
  CK(hipMalloc((void **)&src, bytes)); CK(hipMalloc((void **)&out, bytes)); CK(hipMalloc((void **)&done, 64)); CK(hipMalloc((void **)&bad, 8));
  CK(hipMemset(done, 0, 64)); CK(hipMemset(bad, 0, 8));
  auto ctrl = [&](int r) { return (unsigned long long *)peer[r]; };
  auto payload = [&](int r) { return (double2 *)(peer[r] + 4096 + mbox_bytes); };
  const int up = (rank + 1) % n, lo = (rank + n - 1) % n;
  const int nblk = (int)std::min<size_t>(256, (n2 + 255) / 256);

  // ---- (1) ring push ----
  hbarrier(shm, n, rank, 2);
  auto t0 = std::chrono::steady_clock::now();
  for (int it = 1; it <= iters; it++) {
    // source pattern is tagged by (sender, it): the receiver checks every word against its LOWER neighbour's tag
    hipLaunchKernelGGL(k_fill, dim3(nblk), dim3(256), 0, st, src, n2, (double)(rank * 1000000 + it));
    hipLaunchKernelGGL(k_push, dim3(nblk), dim3(256), 0, st, src, payload(up), n2, ctrl(up) + 0, (unsigned long long)it, done, ctrl(rank) + 1, ctrl(rank) + 2);
    hipLaunchKernelGGL(k_wait_check, dim3(nblk), dim3(256), 0, st, payload(rank), out, n2, ctrl(rank) + 0, (unsigned long long)it,
                       (double)(lo * 1000000 + it), ctrl(rank) + 2, bad, done + 8, ctrl(lo) + 1);
  }
  CK(hipStreamSynchronize(st));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  unsigned long long hbad = 0, herr = 0;
  CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&herr, ctrl(rank) + 2, 8, hipMemcpyDeviceToHost));
  printf("[%d] ring push %zu bytes x %d: %.1f us/round, bad words %llu, poll timeouts %llu\n", rank, bytes, iters, us, hbad, herr);
  fflush(stdout);
  int rc = (hbad || herr) ? 8 : 0;

  // ---- (2) mailbox all-reduce ----
  {
    double *x; CK(hipMalloc((void **)&x, 2048 * 8));
    double **d_mbox; unsigned long long **d_mflag;
    CK(hipMalloc((void **)&d_mbox, MAXR * 8)); CK(hipMalloc((void **)&d_mflag, MAXR * 8));
    double *hm[MAXR]; unsigned long long *hf[MAXR];
    for (int r = 0; r < n; r++) { hm[r] = (double *)(peer[r] + 4096); hf[r] = ctrl(r) + 8; }
    CK(hipMemcpy(d_mbox, hm, n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mflag, hf, n * 8, hipMemcpyHostToDevice));
    for (int nel : {1, 2048}) {
      std::vector<double> h(nel);
      for (int i = 0; i < nel; i++) h[i] = (rank + 1) * 1.0 + i * 0.5;
      hbarrier(shm, n, rank, 50 + (nel > 1));
      static unsigned long long seq = 0;
      const int reps = 2000;
      CK(hipMemcpy(x, h.data(), nel * 8, hipMemcpyHostToDevice));
      auto t1 = std::chrono::steady_clock::now();
      for (int k = 0; k < reps; k++) {
        seq++;
        hipLaunchKernelGGL(k_allreduce, dim3(1), dim3(256), 0, st, x, nel, n, rank, d_mbox, d_mflag, (double *)(arena + 4096), ctrl(rank) + 8, seq, 2048, ctrl(rank) + 2);
        if (k == 0) { CK(hipStreamSynchronize(st)); CK(hipMemcpy(h.data(), x, nel * 8, hipMemcpyDeviceToHost)); }
      }
      CK(hipStreamSynchronize(st));
      double usr = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / reps;
      double want0 = n * (n + 1) / 2.0, wantl = n * (n + 1) / 2.0 + n * (nel - 1) * 0.5;
      CK(hipMemcpy(&herr, ctrl(rank) + 2, 8, hipMemcpyDeviceToHost));
      printf("[%d] mailbox all-reduce n=%d: first result %.3f/%.3f (want %.3f/%.3f), %.2f us each, timeouts %llu\n", rank, nel, h[0], h[nel - 1], want0, wantl, usr, herr);
      if (h[0] != want0 || h[nel - 1] != wantl || herr) rc = 9;
    }
  }
  fflush(stdout);

  // ---- (3) stream write / wait value ----
  {
    hbarrier(shm, n, rank, 55);
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    hipError_t ew = hipStreamWriteValue64(st, ctrl(up) + 4, 7ULL, 0);
    hipError_t es = hipStreamSynchronize(st);
    hbarrier(shm, n, rank, 56);
    hipError_t ewt = hipStreamWaitValue64(st, ctrl(rank) + 4, 7ULL, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFULL);
    hipError_t es2 = (ewt == hipSuccess) ? hipStreamSynchronize(st) : hipSuccess;
    printf("[%d] canUseStreamWaitValue %d; WriteValue64(peer arena): %s / sync %s; WaitValue64(own arena): %s / sync %s\n", rank, can,
           hipGetErrorString(ew), hipGetErrorString(es), hipGetErrorString(ewt), hipGetErrorString(es2));
    (void)hipGetLastError();
    unsigned long long *sig = nullptr;
    hipError_t em = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
    hipIpcMemHandle_t hh;
    hipError_t ei = (em == hipSuccess) ? hipIpcGetMemHandle(&hh, sig) : em;
    printf("[%d] signal memory: alloc %s, ipc export %s\n", rank, hipGetErrorString(em), hipGetErrorString(ei));
    (void)hipGetLastError();
  }
  fflush(stdout);
  hbarrier(shm, n, rank, 60);
  for (int r = 0; r < n; r++) if (r != rank) (void)hipIpcCloseMemHandle(peer[r]);
  hbarrier(shm, n, rank, 61);
  (void)hipFree(arena);
  return rc;
}

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 200;
  size_t bytes = argc > 3 ? (size_t)atol(argv[3]) : (size_t)1 << 20;
  int memkind = argc > 4 ? atoi(argv[4]) : 0;
  const int regrow_mode = argc > 5 ? atoi(argv[5]) : 0;
  if (n < 1 || n > MAXR) return 2;
  char name[64];
  snprintf(name, sizeof name, "/qexprobe_%d", (int)getpid());
  int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) { perror("shm"); return 2; }
  Shm *shm = (Shm *)mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  memset((void *)shm, 0, sizeof(Shm));
  pid_t pids[MAXR];
  for (int r = 0; r < n; r++) {
    pids[r] = fork();
    if (pids[r] == 0) { int rc = regrow_mode ? regrow(r, n, regrow_mode, shm) : child(r, n, iters, bytes, memkind, shm); fflush(stdout); _exit(rc); }
  }
  int worst = 0;
  for (int r = 0; r < n; r++) { int stt = 0; waitpid(pids[r], &stt, 0); int rc = WIFEXITED(stt) ? WEXITSTATUS(stt) : 100 + WTERMSIG(stt); if (rc > worst) worst = rc; }
  shm_unlink(name);
  printf("ipc_probe n=%d kind=%d bytes=%zu: exit %d\n", n, memkind, bytes, worst);
  return worst;
}
