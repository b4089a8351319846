// The reference's stream of standard normals, produced on the host cores in parallel.
//
// The reference draws the Monte-Carlo entropy's eps with np.random.randn (entmc_vbmc.py:67): NumPy's
// legacy global RandomState, i.e. MT19937 words -> 53-bit uniforms -> Marsaglia's polar method.  The
// drop-in default (rng="numpy") reproduces those values draw for draw, and np.random.randn itself is
// the cost of that mode: ~10 ns per normal on one core, 48 ms for the 5e6 of config 3 against a
// 0.1 ms evaluation.  The stream is sequential only in the MT19937 recurrence (cheap integer work,
// one 624-word block from the one before); everything expensive is a pure function of the word
// position: attempt t of the polar method always consumes words [4t, 4t+4) -- two doubles of two
// words each -- whether it is accepted or not, and an accepted attempt always yields two values,
// (f*x2, f*x1) in that order.  So: one thread runs the recurrence, all threads evaluate attempts, the
// accepted pairs are placed by a prefix sum over per-range counts.  The arithmetic is NumPy's
// expression for expression (same libm log, IEEE sqrt and division, no contraction), so the values
// are bit-identical; tests/test_host_randn.py compares them and the state left behind with
// np.random.randn on every path (cached second value, odd counts, block boundaries).
//
// This is host logic of the "numpy" draw source, not a stand-in for a kernel: the draws still go to
// the device with vbmc_set_eps and every density is evaluated there.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/vbmc_hip.h"

#ifndef __HIP_DEVICE_COMPILE__  // host code only (x86 dispatch builtins): nothing here for the gfx950 pass

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline uint32_t twist(uint32_t hi, uint32_t lo) {
  const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
  return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

// d = the state after one regeneration of s (NumPy's mt19937_gen, written out of place so that each
// of the three runs is a loop without a carried dependence shorter than 227 elements: the compiler
// vectorises them).  Integer work only, so the 8-wide AVX2 build of the same source gives the same
// words; it is chosen once at run time when the CPU has it (a 16-wide AVX-512 build was measured on
// the MI355X box's EPYC 9575F: 2.07 ms for the pass over the recurrence against 1.50 ms, not kept).
#define VBMC_MT_BLOCK_BODY                                                                                   \
  for (int i = 0; i < MT_N - MT_M; ++i) d[i] = s[i + MT_M] ^ twist(s[i], s[i + 1]);                           \
  for (int i = MT_N - MT_M; i < 2 * (MT_N - MT_M); ++i) d[i] = d[i - (MT_N - MT_M)] ^ twist(s[i], s[i + 1]);  \
  for (int i = 2 * (MT_N - MT_M); i < MT_N - 1; ++i) d[i] = d[i - (MT_N - MT_M)] ^ twist(s[i], s[i + 1]);     \
  d[MT_N - 1] = d[MT_M - 1] ^ twist(s[MT_N - 1], d[0]);
void mt_next_block_base(const uint32_t* __restrict__ s, uint32_t* __restrict__ d) { VBMC_MT_BLOCK_BODY }
__attribute__((target("avx2"))) void mt_next_block_avx2(const uint32_t* __restrict__ s, uint32_t* __restrict__ d) {
  VBMC_MT_BLOCK_BODY
}
#undef VBMC_MT_BLOCK_BODY
using BlockFn = void (*)(const uint32_t*, uint32_t*);
const BlockFn mt_next_block = [] {
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx2") ? (BlockFn)mt_next_block_avx2 : (BlockFn)mt_next_block_base;
}();

inline uint32_t temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// mt19937_next_double of two consecutive words
inline double to_double(uint32_t w0, uint32_t w1) {
  const int32_t a = (int32_t)(temper(w0) >> 5), b = (int32_t)(temper(w1) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

struct Attempt {
  double x1, x2, r2;
  bool ok;
};
inline Attempt attempt_at(const uint32_t* u) {  // legacy_gauss's loop body on words u[0..3]
  Attempt a;
  a.x1 = 2.0 * to_double(u[0], u[1]) - 1.0;
  a.x2 = 2.0 * to_double(u[2], u[3]) - 1.0;
  a.r2 = a.x1 * a.x1 + a.x2 * a.x2;
  a.ok = !(a.r2 >= 1.0 || a.r2 == 0.0);
  return a;
}

}  // namespace

// vbmc_mt19937_randn (include/vbmc_hip.h): layout of the work.  Stream word w (w = 0 at key[pos]) is word pos + w of the block sequence
// 0 = the current key, 1 = its successor, ...; attempt t reads stream words [4t, 4t + 4).  The calling
// thread runs the recurrence once over all blocks the request can need WITHOUT storing the stream --
// two blocks ping-pong in its L1, 0.15 ns per word -- and keeps only one checkpoint (a 2.5 KB state)
// per range of attempts.  Each worker regenerates the words of its own range from its checkpoint
// through a small private scratch (a 50 MB stream written by one core and read by all the others was
// measured at 7.5 ms for the writes alone, from the second call on: every line had to be recalled
// from the readers' caches; per-range copies of it cost 3 ms of page faults and unmapping per
// call), counts its accepted attempts, and after the prefix sum over the ranges regenerates them
// once more and writes its values where they belong.
namespace {

struct Range {
  int64_t a0 = 0, a1 = 0;        // attempts
  int64_t blk0 = 0;              // block holding stream word 4 * a0
  int64_t skip = 0;              // position of that word inside block blk0
  uint32_t state[MT_N];          // block blk0 (blk0 = 0: the caller's key)
  int64_t accepted = 0, first_pair = 0, end_attempt = -1;
  double last_second = 0.0;
  uint32_t end_key[MT_N];        // the block holding the last word of end_attempt
};

// f(t, words of attempt t) for t = 0 .. n_att-1 (until f returns false), the words regenerated from
// `state` (the block holding the first attempt's first word, at offset `skip`) through a scratch of
// C + 1 blocks that stays in the core's L1/L2; `block_of_last` receives the block that holds the last
// word handed to f.
template <class F> void walk_attempts(const uint32_t* state, int64_t skip, int64_t n_att, uint32_t* block_of_last, F&& f) {
  constexpr int C = 24;
  uint32_t buf[(C + 1) * MT_N];
  std::memcpy(buf, state, sizeof(uint32_t) * MT_N);
  int64_t have = MT_N, off = skip, t = 0;
  while (t < n_att) {
    for (; have < (C + 1) * MT_N; have += MT_N) mt_next_block(buf + have - MT_N, buf + have);
    for (; t < n_att && off + 4 <= have; off += 4, ++t)
      if (!f(t, buf + off)) {
        if (block_of_last) std::memcpy(block_of_last, buf + (off + 3) / MT_N * MT_N, sizeof(uint32_t) * MT_N);
        return;
      }
    // slide: the last block (which holds the next attempt's first word) moves to the front
    const int64_t keep = have - MT_N;
    std::memmove(buf, buf + keep, sizeof(uint32_t) * MT_N);
    off -= keep;
    have = MT_N;
  }
}

}  // namespace

// progress(user, m): out[0 .. m) is final (called from the calling thread while the team still writes
// the rest; m grows; the caller handles whatever is left after the return)
using RandnProgress = void (*)(void* user, int64_t m);
static int randn_impl(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, int64_t n, int n_threads,
                      RandnProgress progress, void* user);

extern "C" int vbmc_mt19937_randn(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, int64_t n,
                                  int n_threads) {
  // no exception crosses the C boundary: every allocation below happens before the first thread is
  // started, so an allocation failure unwinds with nothing running (thread creation failures are
  // handled where they occur)
  try {
    return randn_impl(key, pos, has_gauss, gauss, out, n, n_threads, nullptr, nullptr);
  } catch (const std::exception&) {
    return VBMC_E_NOMEM;
  }
}

// the same with the progress callback (declared in common.h; vbmc_set_eps_numpy uploads the finished
// prefix of the draws while the rest is generated)
int randn_with_progress(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, int64_t n, int n_threads,
                        void (*progress)(void*, int64_t), void* user) {
  try {
    return randn_impl(key, pos, has_gauss, gauss, out, n, n_threads, progress, user);
  } catch (const std::exception&) {
    return VBMC_E_NOMEM;
  }
}

static int randn_impl(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, int64_t n, int n_threads,
                      RandnProgress progress, void* user) {
  if (!key || !pos || !has_gauss || !gauss || n < 0 || (n > 0 && !out) || *pos < 0 || *pos > MT_N) return VBMC_E_ARG;
  int64_t produced = 0;
  if (n > 0 && *has_gauss) {  // the second value of the last accepted attempt comes first
    out[produced++] = *gauss;
    *has_gauss = 0;
    *gauss = 0.0;
  }
  const int64_t rest = n - produced;
  if (rest == 0) return VBMC_OK;
  const int64_t pairs = (rest + 1) / 2;  // accepted attempts to consume; the last gives one value when `rest` is odd
  if (n_threads <= 0) n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
  if (pairs < 16384) n_threads = 1;
  static const bool debug = getenv("VBMC_RANDN_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const int pos0 = *pos;
  double* const dst0 = out + produced;

  // attempts are accepted with probability pi/4: take 6 sigma more than the expected number, and
  // start over with a wider margin in the (practically never taken) case that it was not enough
  for (double sigmas = 6.0;; sigmas *= 4.0) {
    const double t0 = now();
    const double expect = (double)pairs / 0.7853981633974483;
    const int64_t attempts = (int64_t)(expect + sigmas * std::sqrt(expect) + 64.0);
    const int T = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, attempts / 4096));
    // several ranges per thread, handed out in order: a range can be counted as soon as the
    // recurrence has passed its first block, so the counting hides behind the checkpoint pass
    const int R = T == 1 ? 1 : 8 * T;
    std::vector<Range> rg(R);
    for (int r = 0; r < R; ++r) {
      rg[r].a0 = attempts * r / R;
      rg[r].a1 = attempts * (r + 1) / R;
      const int64_t w = pos0 + 4 * rg[r].a0;  // word index in the block sequence
      rg[r].blk0 = w / MT_N;
      rg[r].skip = w - rg[r].blk0 * MT_N;
    }
    struct alignas(64) Flag {  // one cache line each: a waiting worker polls its own flag only
      std::atomic<int> v{0};
    };
    std::unique_ptr<Flag[]> ready(new Flag[R]);
    std::atomic<int> next_count{0}, next_value{0}, counted{0}, phase2{0};
    static const bool backoff = !getenv("VBMC_RANDN_NOSLEEP");  // experiments
    auto wait_for = [](auto&& cond) {  // short waits spin, long ones get out of the way of the threads that work
      for (int spins = 0; !cond(); ++spins) {
        if (spins > 2000 && backoff) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else if (spins > 64) std::this_thread::yield();
      }
    };
    auto count_range = [&](int r) {
      Range& g = rg[r];
      int64_t c = 0;
      walk_attempts(g.state, g.skip, g.a1 - g.a0, nullptr, [&](int64_t, const uint32_t* u) {
        c += attempt_at(u).ok ? 1 : 0;
        return true;
      });
      g.accepted = c;
    };
    // the values (words regenerated once more: 0.2 ns each against ~10 ns for an accepted attempt),
    // placed by the prefix sums; the range holding the final pair stops there
    auto value_range = [&](int r) {
      Range& g = rg[r];
      int64_t p = g.first_pair;  // index of the next accepted pair
      if (p >= pairs) return;
      walk_attempts(g.state, g.skip, g.a1 - g.a0, g.end_key, [&](int64_t t, const uint32_t* u) {
        const Attempt a = attempt_at(u);
        if (!a.ok) return true;
        const double f = std::sqrt(-2.0 * std::log(a.r2) / a.r2);
        const int64_t o = 2 * p;
        dst0[o] = f * a.x2;
        if (o + 1 < rest) dst0[o + 1] = f * a.x1;
        else g.last_second = f * a.x1;
        if (++p < pairs) return true;
        g.end_attempt = g.a0 + t;
        return false;
      });
    };
    auto worker = [&] {
      for (int r; (r = next_count.fetch_add(1, std::memory_order_relaxed)) < R;) {
        wait_for([&] { return ready[r].v.load(std::memory_order_acquire) != 0; });
        count_range(r);
        counted.fetch_add(1, std::memory_order_release);
      }
      wait_for([&] { return phase2.load(std::memory_order_acquire) != 0; });
      if (phase2.load(std::memory_order_relaxed) < 0) return;
      for (int r; (r = next_value.fetch_add(1, std::memory_order_relaxed)) < R;) {
        value_range(r);
        ready[r].v.store(2, std::memory_order_release);  // this range's values are in place
      }
    };
    // the team is started by its first member (creating 64 threads takes the calling thread ~1.5 ms,
    // as long as its pass over the recurrence)
    std::vector<std::thread> team;
    std::thread starter;
    if (T > 1) {
      team.reserve(T);
      auto start_team = [&] {
        try {
          for (int i = 1; i < T; ++i) team.emplace_back(worker);
        } catch (const std::system_error&) {
          // fewer threads than asked for: the ranges are handed out dynamically, whoever exists takes them
        }
        worker();
      };
      try {
        starter = std::thread(start_team);
      } catch (const std::system_error&) {
        // no thread at all: the calling thread does everything below
      }
    }
    const bool alone = !starter.joinable();
    {  // checkpoints: one pass of the recurrence, nothing stored but the ranges' first blocks
      uint32_t buf[2][MT_N];
      std::memcpy(buf[0], key, sizeof(buf[0]));
      int cur = 0;
      int64_t blk = 0;
      for (int r = 0; r < R; ++r) {
        for (; blk < rg[r].blk0; ++blk, cur ^= 1) mt_next_block(buf[cur], buf[cur ^ 1]);
        std::memcpy(rg[r].state, buf[cur], sizeof(buf[0]));
        ready[r].v.store(1, std::memory_order_release);
      }
    }
    const double t1 = now();
    if (alone)
      for (int r = 0; r < R; ++r) count_range(r);
    else
      wait_for([&] { return counted.load(std::memory_order_acquire) == R; });
    int64_t total = 0;
    for (int r = 0; r < R; ++r) {
      rg[r].first_pair = total;
      total += rg[r].accepted;
    }
    const double t2 = now();
    phase2.store(total < pairs ? -1 : 1, std::memory_order_release);
    if (alone && total >= pairs)
      for (int r = 0; r < R; ++r) value_range(r);
    if (!alone && total >= pairs && progress) {
      // hand the finished prefix of the output to the caller while the rest is being written (the
      // upload of the draws then runs beside their generation): ranges are taken in order
      int64_t reported = 0;
      for (int r = 0; r < R;) {
        wait_for([&] { return ready[r].v.load(std::memory_order_acquire) == 2; });
        while (r < R && ready[r].v.load(std::memory_order_acquire) == 2) ++r;
        const int64_t pairs_done = r < R ? std::min(rg[r].first_pair, pairs) : pairs;
        const int64_t m = produced + std::min(2 * pairs_done, rest);
        if (m - reported >= ((int64_t)1 << 19) || r == R) {  // >= 4 MB at a time
          progress(user, m);
          reported = m;
        }
      }
    }
    if (starter.joinable()) starter.join();  // (its team is complete once it has returned)
    for (auto& th : team) th.join();
    if (total < pairs) continue;
    if (debug)
      fprintf(stderr, "[vbmc] randn: checkpoints %.2f, count done %.2f later, values %.2f ms (%d threads)\n", t1 - t0,
              t2 - t1, now() - t2, T);
    for (int r = 0; r < R; ++r) {
      const Range& g = rg[r];
      if (g.end_attempt < 0) continue;
      if (rest & 1) {
        *has_gauss = 1;
        *gauss = g.last_second;
      }
      // the state NumPy would be left in: right behind the words of the final attempt
      const int64_t w = pos0 + 4 * (g.end_attempt + 1);  // word index in the block sequence, >= 4
      const int64_t b = (w - 1) / MT_N;                  // the block holding the last word read
      if (b > 0) std::memcpy(key, g.end_key, sizeof(uint32_t) * MT_N);
      *pos = (int)(w - b * MT_N);
      return VBMC_OK;
    }
    return VBMC_E_ARG;  // unreachable: total >= pairs means some range completed the request
  }
}

#endif  // !__HIP_DEVICE_COMPILE__
