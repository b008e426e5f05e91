// test_peer_shm.cpp -- the host rendezvous of the peer transport (qex_amd/csrc/peer_shm.cpp), CPU only.
// What QMP gives QEX (src/comms/commsQmp.nim:14-33 init, :127-140 barrier / max) restated over a POSIX shm segment:
// N forked processes must (1) meet, (2) pass 1000 barriers without one rank ever running a generation ahead,
// (3) agree bit for bit on max / min / rank-ordered sum, (4) get an error -- not a hang -- when a rank never arrives or
// reports a failure, (5) refuse a slot that is already taken (a unique id serves one comm_init), and (6) take ONE transport
// decision for the job in comm_init's `auto` mode (peer_host_choose) -- including jobs that span nodes, whose ranks never share
// a segment: RCCL by the launcher's hint without any wait, RCCL after the rendezvous timeout without one, never an error.
#include "../../qex_amd/csrc/peer_shm.h"
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <initializer_list>
#include <sys/wait.h>
#include <unistd.h>

static char g_err[512];
void qexhip_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

static int run_ranks(int n, int (*fn)(int, int, const unsigned char *), int salt) {
  unsigned char id[128];
  // the id is made in the parent so that every child holds the same one
  for (int i = 0; i < 128; i++) id[i] = (unsigned char)(i * 7 + salt);
  const int pid = (int)getpid() ^ (salt << 16);
  memcpy(id, &pid, sizeof pid);
  pid_t pids[PEER_MAXR];
  for (int r = 0; r < n; r++) {
    pids[r] = fork();
    if (pids[r] == 0) _exit(fn(n, r, id));
  }
  int worst = 0;
  for (int r = 0; r < n; r++) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 100;
    if (rc > worst) worst = rc;
  }
  return worst;
}

static int t_collectives(int n, int rank, const unsigned char *id) {
  PeerHost h;
  if (peer_host_open(&h, id, n, rank, 30.0)) { fprintf(stderr, "[%d] open: %s\n", rank, g_err); return 1; }
  for (int k = 0; k < 1000; k++) {
    if (peer_host_barrier(&h)) { fprintf(stderr, "[%d] barrier %d: %s\n", rank, k, g_err); return 2; }
    // nobody is more than one generation away from anybody else after a barrier
    for (int r = 0; r < n; r++) {
      const long g = h.shm->s[r].gen;
      if (g < h.gen || g > h.gen + 1) { fprintf(stderr, "[%d] generation skew %ld vs %ld\n", rank, g, h.gen); return 3; }
    }
    if (k == 0) peer_host_unlink(&h);       // the name can go once everybody has mapped it
  }
  for (int k = 0; k < 200; k++) {
    double v[4] = {rank + 0.25 * k, -(double)rank, 1.0 / (rank + 1 + k), (rank == k % n) ? NAN : 1.0};
    double mx[4], mn[4], sm[4];
    memcpy(mx, v, sizeof v); memcpy(mn, v, sizeof v); memcpy(sm, v, sizeof v);
    if (peer_host_allreduce(&h, mx, 4, 0) || peer_host_allreduce(&h, mn, 4, 1) || peer_host_allreduce(&h, sm, 3, 2)) { fprintf(stderr, "[%d] allreduce: %s\n", rank, g_err); return 4; }
    double s2 = 0;
    for (int r = 0; r < n; r++) s2 += 1.0 / (r + 1 + k);                    // rank order: must match bit for bit
    if (mx[0] != n - 1 + 0.25 * k || mx[1] != 0.0 || mn[0] != 0.25 * k || mn[1] != -(double)(n - 1) || sm[2] != s2 || !std::isnan(mx[3]) || !std::isnan(mn[3])) {
      fprintf(stderr, "[%d] wrong reduction at %d: %g %g %g %g %.17g vs %.17g %g\n", rank, k, mx[0], mx[1], mn[0], mn[1], sm[2], s2, mx[3]);
      return 5;
    }
  }
  peer_host_close(&h);
  return 0;
}

// rank n-1 never arrives: the others must come back with an error within the timeout
static int t_missing(int n, int rank, const unsigned char *id) {
  if (rank == n - 1) return 0;
  PeerHost h;
  if (peer_host_open(&h, id, n, rank, 1.0)) return 1;
  const auto t0 = std::chrono::steady_clock::now();
  const int e = peer_host_barrier(&h);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rank == 0) peer_host_unlink(&h);
  peer_host_close(&h);
  if (e == 0 || dt > 10.0 || !strstr(g_err, "peer rendezvous")) { fprintf(stderr, "[%d] missing rank: rc %d after %.1f s (%s)\n", rank, e, dt, g_err); return 2; }
  return 0;
}

// rank 0 reports a failure instead of arriving: the others fail fast, long before the timeout
static int t_failed(int n, int rank, const unsigned char *id) {
  PeerHost h;
  if (peer_host_open(&h, id, n, rank, 60.0)) return 1;
  if (peer_host_barrier(&h)) return 2;
  if (rank == 0) { peer_host_unlink(&h); peer_host_fail(&h); peer_host_close(&h); return 0; }
  const auto t0 = std::chrono::steady_clock::now();
  const int e = peer_host_barrier(&h);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  peer_host_close(&h);
  if (e == 0 || dt > 5.0) { fprintf(stderr, "[%d] failed rank: rc %d after %.1f s\n", rank, e, dt); return 3; }
  return 0;
}

// ---- the transport decision (peer_host_choose): mode 0 rccl, 2 peer, 3 rccl + mailbox sums ----
static int g_wish = 0, g_expect_mode = 0, g_expect_err = 0, g_same_bus = 0, g_present = 0;
static double g_max_s = 0;
static int t_choose(int n, int rank, const unsigned char *id) {
  if (g_present && rank >= g_present) return 0;            // "on another node": never opens THIS node's segment
  PeerHost h;
  char bus[32];
  snprintf(bus, sizeof bus, "0000:%02x:00.0", g_same_bus ? 7 : 16 + rank);
  int mode = -1, shared = -1;
  const auto t0 = std::chrono::steady_clock::now();
  const int e = peer_host_choose(&h, id, n, rank, g_wish, rank, bus, 1.5, &mode, &shared);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!e && (mode == 2 || mode == 3)) {                    // the segment stays open for the handle exchange: one more barrier, then close
    if (peer_host_barrier(&h)) return 5;
    peer_host_close(&h);
  }
  if ((e != 0) != (g_expect_err != 0) || (!e && mode != g_expect_mode) || dt > g_max_s || (!e && mode == 2 && shared != g_same_bus)) {
    fprintf(stderr, "[%d] choose: rc %d mode %d shared %d after %.2f s (want err %d mode %d within %.1f s): %s\n", rank, e, mode, shared, dt,
            g_expect_err, g_expect_mode, g_max_s, g_err);
    return 6;
  }
  return 0;
}
static int choose_case(const char *what, int n, int present, int wish, int same_bus, const char *hint, int expect_mode, int expect_err, double max_s, int salt) {
  g_wish = wish; g_expect_mode = expect_mode; g_expect_err = expect_err; g_same_bus = same_bus; g_present = present; g_max_s = max_s;
  if (hint) setenv("QEXHIP_LOCAL_RANKS", hint, 1); else unsetenv("QEXHIP_LOCAL_RANKS");
  unsetenv("LOCAL_WORLD_SIZE");
  const int rc = run_ranks(n, t_choose, salt);
  unsetenv("QEXHIP_LOCAL_RANKS");
  printf("transport decision, %s: %s\n", what, rc ? "FAILED" : "ok");
  return rc;
}

int main() {
  int bad = 0;
  bad |= choose_case("4 ranks on 4 devices of one node -> rccl + mailbox sums", 4, 0, 0, 0, nullptr, 3, 0, 1.0, 50);
  bad |= choose_case("4 ranks sharing one device -> peer", 4, 0, 0, 1, nullptr, 2, 0, 1.0, 51);
  bad |= choose_case("launcher says 2 of 4 ranks are local -> rccl at once, no segment", 4, 2, 0, 0, "2", 0, 0, 0.2, 52);
  bad |= choose_case("2 of 4 ranks on this node, no hint -> every local rank times out alike -> rccl", 4, 2, 0, 0, nullptr, 0, 0, 4.0, 53);
  bad |= choose_case("the same with an explicit wish for peer -> an error, not a fallback", 4, 2, 2, 0, nullptr, 0, 1, 4.0, 54);
  bad |= choose_case("wish rccl -> no rendezvous", 4, 0, 1, 0, nullptr, 0, 0, 0.2, 55);
  {
    // after a failed rendezvous no name is left behind for a retry with the same id to trip over (whoever times out unlinks)
    unsigned char id[128];
    for (int i = 0; i < 128; i++) id[i] = (unsigned char)(3 * i + 1);
    const int pid = (int)getpid();
    memcpy(id + 16, &pid, sizeof pid);
    PeerHost a;
    int mode = -1, shared = -1;
    const int e1 = peer_host_choose(&a, id, 2, 1, 2, 0, "bus", 0.5, &mode, &shared);      // rank 1 alone, insists on peer: times out
    PeerHost b;
    const int e2 = peer_host_open(&b, id, 2, 1, 0.5);                                      // the retry finds slot 1 free again
    if (!e2) { peer_host_unlink(&b); peer_host_close(&b); }
    printf("a failed rendezvous leaves no stale slot: %s\n", (e1 != 0 && e2 == 0) ? "ok" : "FAILED");
    bad |= !(e1 != 0 && e2 == 0);
  }
  for (int n : {1, 2, 4, 8}) {
    const int rc = run_ranks(n, t_collectives, 10 + n);
    printf("collectives, %d ranks: %s\n", n, rc ? "FAILED" : "ok");
    bad |= rc;
  }
  { const int rc = run_ranks(3, t_missing, 40); printf("a rank that never arrives: %s\n", rc ? "FAILED" : "ok"); bad |= rc; }
  { const int rc = run_ranks(3, t_failed, 41); printf("a rank that reports a failure: %s\n", rc ? "FAILED" : "ok"); bad |= rc; }
  {
    // a slot can be taken once
    unsigned char id[128];
    for (int i = 0; i < 128; i++) id[i] = (unsigned char)(255 - i);
    const int pid = (int)getpid();
    memcpy(id + 8, &pid, sizeof pid);
    PeerHost a, b;
    const int e1 = peer_host_open(&a, id, 2, 0, 1.0), e2 = peer_host_open(&b, id, 2, 0, 1.0);
    peer_host_close(&a);
    printf("a slot is taken once: %s\n", (e1 == 0 && e2 != 0) ? "ok" : "FAILED");
    bad |= !(e1 == 0 && e2 != 0);
    PeerHost c;
    const int e3 = peer_host_open(&c, id, PEER_MAXR + 1, 0, 1.0);
    bad |= (e3 == 0);
  }
  printf("peer rendezvous: %s\n", bad ? "FAILED" : "Passed");
  return bad ? 1 : 0;
}
