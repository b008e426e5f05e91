// peer_shm.h -- host-side rendezvous of the ranks of ONE node through a POSIX shared-memory segment (no GPU, no HIP).
//
// The peer-memory transport (peer.hip) needs an out-of-band channel before any device memory is shared: to learn which
// device every rank is bound to (transport choice), to pass hipIpc handles around, and for the few collectives whose
// operands live on the host (comm_allreduce_max, the agreement on options).  QEX has QMP for this
// (src/comms/commsQmp.nim:14-33,127-140: QMP_init_msg_passing, QMP_barrier, QMP_max_double); a library behind a C ABI
// that is handed only a 128-byte id has to bring its own.  The segment is named after a hash of that id.
//
// Every wait is bounded (timeout_s): a rank that never arrives turns into QEXHIP_ERR_COMM on the others, not into a hang.
#pragma once
#include <cstddef>
#include <cstdint>

enum { PEER_MAXR = 16, PEER_HANDLE_BYTES = 64, PEER_NHANDLE = 3 };

struct PeerShmSlot {
  volatile long gen;                 // barrier generation this rank has reached
  volatile long seen;                // 1 once the rank has written its identity below
  int pid, device, wish;             // wish: 0 auto, 1 rccl, 2 peer
  char bus[32];                      // PCI bus id of the bound device
  char host[64];
  unsigned char handle[PEER_NHANDLE][PEER_HANDLE_BYTES];   // hipIpcMemHandle_t of [ctrl | arena of stream class 0 | class 1]
  unsigned long long cap[PEER_NHANDLE];                    // bytes behind each handle
  double red[8];                     // operand of a host collective
  volatile int failed;               // this rank gave up (timeout / error): everybody else fails fast
  char pad[44];
};

struct PeerShm {
  volatile long magic;
  PeerShmSlot s[PEER_MAXR];
};

struct PeerHost {                    // one rank's view
  PeerShm *shm = nullptr;
  int nranks = 0, rank = 0;
  long gen = 0;
  double timeout_s = 120.0;
  char name[48] = {0};
  bool unlinked = false;
};

// All return 0 or a negative QEXHIP_ERR_* and set the library's error string.
int peer_host_open(PeerHost *h, const unsigned char id[128], int nranks, int rank, double timeout_s);   // creates / maps the segment (no wait)
int peer_host_barrier(PeerHost *h);                                                                    // bounded
int peer_host_allreduce(PeerHost *h, double *v, int n, int op);                                        // op 0 max, 1 min, 2 sum (rank order); n <= 8; two barriers
void peer_host_fail(PeerHost *h);                                                                      // mark this rank failed (and drop the segment's name: whoever
                                                                                                       // retries with the same id must not find a stale slot)
void peer_host_close(PeerHost *h);                                                                     // unmaps; rank 0 unlinks the name if still there
void peer_host_unlink(PeerHost *h);                                                                    // drop the name early (after the first barrier nobody opens it again)

// What the launcher says about how many of the job's ranks run on THIS node (0: it does not say): QEXHIP_LOCAL_RANKS, then
// torch.distributed.run's LOCAL_WORLD_SIZE, Open MPI, MPICH / Hydra, Slurm.
int peer_host_local_ranks_hint();
// The transport decision of comm_init (comm.cpp), ONE for the whole job.  wish: 0 auto, 1 rccl, 2 peer, 3 rccl + mailbox sums.
// *mode = 0 rccl (nothing left open), 2 peer, 3 rccl + mailbox sums (`h` stays open for the handle exchange).  *shared = two ranks sit on
// one device.  auto never fails for want of a rendezvous: a job that spans nodes (the hint says so, or nobody else shows up in the
// segment within timeout_s -- every node's ranks time out alike) takes rccl; wish 2 / 3 return the error instead.
int peer_host_choose(PeerHost *h, const unsigned char id[128], int nranks, int rank, int wish, int device, const char *bus,
                     double timeout_s, int *mode, int *shared);
