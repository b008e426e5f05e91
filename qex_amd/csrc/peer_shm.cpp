// peer_shm.cpp -- see peer_shm.h.  Pure host code: POSIX shm + C11-style atomics, no HIP call.
#include "peer_shm.h"
#include "../../include/qexhip.h"
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

void qexhip_set_error(const char *fmt, ...);

static unsigned long long fnv1a(const unsigned char *p, size_t n) {
  unsigned long long h = 1469598103934665603ULL;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ULL; }
  return h;
}

int peer_host_open(PeerHost *h, const unsigned char id[128], int nranks, int rank, double timeout_s) {
  if (!h || !id || nranks < 1 || nranks > PEER_MAXR || rank < 0 || rank >= nranks) {
    qexhip_set_error("peer rendezvous: nranks %d / rank %d out of range (at most %d ranks per node)", nranks, rank, PEER_MAXR);
    return QEXHIP_ERR_ARG;
  }
  h->nranks = nranks; h->rank = rank; h->gen = 0; h->timeout_s = timeout_s; h->unlinked = false;
  snprintf(h->name, sizeof h->name, "/qexhip_%016llx", fnv1a(id, 128));
  int fd = shm_open(h->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { qexhip_set_error("peer rendezvous: shm_open(%s): %s", h->name, strerror(errno)); return QEXHIP_ERR_COMM; }
  // every rank sizes the segment (idempotent); a fresh segment reads as zeros, which is the initial state of every field
  if (ftruncate(fd, sizeof(PeerShm)) != 0) { qexhip_set_error("peer rendezvous: ftruncate: %s", strerror(errno)); close(fd); return QEXHIP_ERR_COMM; }
  void *p = mmap(nullptr, sizeof(PeerShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { qexhip_set_error("peer rendezvous: mmap: %s", strerror(errno)); return QEXHIP_ERR_COMM; }
  h->shm = (PeerShm *)p;
  PeerShmSlot &me = h->shm->s[rank];
  if (__atomic_exchange_n(&me.seen, 1L, __ATOMIC_ACQ_REL) != 0) {
    // the id was used before by a job that is still around (or died without closing): never join a stale segment
    qexhip_set_error("peer rendezvous: slot %d of %s is already taken -- a unique id must be used for one comm_init only", rank, h->name);
    munmap(p, sizeof(PeerShm)); h->shm = nullptr;
    return QEXHIP_ERR_COMM;
  }
  me.pid = (int)getpid();
  me.failed = 0;
  if (gethostname(me.host, sizeof me.host) != 0) me.host[0] = 0;
  me.host[sizeof me.host - 1] = 0;
  return 0;
}

void peer_host_fail(PeerHost *h) {
  if (!h || !h->shm) return;
  __atomic_store_n(&h->shm->s[h->rank].failed, 1, __ATOMIC_RELEASE);
  // Whoever notices a failure drops the NAME (unlinking while others have the segment mapped is safe): with rank 0 the absent or late
  // one nobody else would, and a retry of comm_init with the same id would find its slot "already taken".
  peer_host_unlink(h);
}

int peer_host_barrier(PeerHost *h) {
  if (!h || !h->shm) return QEXHIP_ERR_STATE;
  const long my = ++h->gen;
  __atomic_store_n(&h->shm->s[h->rank].gen, my, __ATOMIC_RELEASE);
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  for (int r = 0; r < h->nranks; r++) {
    while (__atomic_load_n(&h->shm->s[r].gen, __ATOMIC_ACQUIRE) < my) {
      if (__atomic_load_n(&h->shm->s[r].failed, __ATOMIC_ACQUIRE)) {
        qexhip_set_error("peer rendezvous: rank %d reported a failure", r);
        peer_host_fail(h);
        return QEXHIP_ERR_COMM;
      }
      if (++spins > 2000) {       // ~ the first 100 us busy, then yield the core
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > h->timeout_s) {
          qexhip_set_error("peer rendezvous: rank %d did not reach barrier %ld within %.0f s (all ranks of a peer-transport job must "
                           "run on one node; set QEXHIP_TRANSPORT=rccl for anything else)", r, my, h->timeout_s);
          peer_host_fail(h);
          return QEXHIP_ERR_COMM;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(dt < 0.01 ? 5 : 100));
      }
    }
  }
  return 0;
}

int peer_host_allreduce(PeerHost *h, double *v, int n, int op) {
  if (!h || !h->shm || n < 0 || n > 8) return QEXHIP_ERR_ARG;
  PeerShmSlot &me = h->shm->s[h->rank];
  for (int i = 0; i < n; i++) me.red[i] = v[i];
  if (int e = peer_host_barrier(h)) return e;
  for (int i = 0; i < n; i++) {
    double acc = h->shm->s[0].red[i];
    for (int r = 1; r < h->nranks; r++) {
      const double x = h->shm->s[r].red[i];
      if (op == 0) acc = (x > acc || x != x) ? x : acc;       // a NaN wins on every rank alike
      else if (op == 1) acc = (x < acc || x != x) ? x : acc;
      else acc += x;                                          // rank order: the same bits everywhere
    }
    v[i] = acc;
  }
  return peer_host_barrier(h);     // nobody overwrites its operand before everybody has read it
}

void peer_host_unlink(PeerHost *h) {
  if (h && h->name[0] && !h->unlinked) { (void)shm_unlink(h->name); h->unlinked = true; }
}

void peer_host_close(PeerHost *h) {
  if (!h) return;
  if (h->shm) {
    if (h->rank == 0) peer_host_unlink(h);
    munmap((void *)h->shm, sizeof(PeerShm));
    h->shm = nullptr;
  }
}

int peer_host_local_ranks_hint() {
  for (const char *name : {"QEXHIP_LOCAL_RANKS", "LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE"}) {
    const char *e = getenv(name);
    if (e && *e) { const int v = atoi(e); if (v > 0) return v; }
  }
  return 0;
}

int peer_host_choose(PeerHost *h, const unsigned char id[128], int nranks, int rank, int wish, int device, const char *bus,
                     double timeout_s, int *mode, int *shared) {
  *mode = 0; *shared = 0;
  if (wish == 1 || nranks > PEER_MAXR) return 0;
  if (wish == 0 && nranks == 1) return 0;          // one rank, no preference: nobody to meet (the one-rank RCCL communicator of the rehearsals)
  if (wish == 0) {
    const int local = peer_host_local_ranks_hint();
    if (local > 0 && local < nranks) return 0;      // the job spans nodes: RCCL, without waiting for a rendezvous that cannot complete
  }
  if (int e = peer_host_open(h, id, nranks, rank, timeout_s)) {
    if (wish != 0) return e;
    // auto: no shared-memory segment to meet in (no /dev/shm, a sandbox): RCCL is the only transport left, and says so itself if the
    // ranks turn out to share a device.  (Ranks that DID open the segment run into the barrier's timeout below and follow.)
    fprintf(stderr, "libqexhip: rank %d: no rendezvous segment: taking the RCCL transport\n", rank);
    return 0;
  }
  PeerShmSlot &me = h->shm->s[rank];
  me.device = device;
  me.wish = wish;
  snprintf(me.bus, sizeof me.bus, "%s", bus ? bus : "");
  if (int e = peer_host_barrier(h)) {
    peer_host_close(h);
    if (wish != 0) return e;
    // auto: somebody never arrived.  Ranks on other nodes cannot (and no launcher variable told us): every node's ranks time out alike --
    // a failed barrier fails for all its participants -- so "RCCL" is still ONE decision for the whole job.  If a rank is really
    // gone, RCCL's own bootstrap says so next.
    fprintf(stderr, "libqexhip: rank %d: the node-local rendezvous did not complete within %.0f s: taking the RCCL transport\n", rank, timeout_s);
    return 0;
  }
  int any[4] = {0, 0, 0, 0};
  for (int r = 0; r < nranks; r++) {
    const PeerShmSlot &a = h->shm->s[r];
    if (a.wish >= 0 && a.wish <= 3) any[a.wish] = 1;
    for (int q = 0; q < r; q++) {
      const PeerShmSlot &b = h->shm->s[q];
      if (!strncmp(a.bus, b.bus, sizeof a.bus) && !strncmp(a.host, b.host, sizeof a.host)) *shared = 1;
    }
  }
  *mode = (any[2] || *shared) ? 2 : 3;             // (distinct devices, nobody insists on peer: the mailboxes ride along with RCCL)
  return 0;
}
