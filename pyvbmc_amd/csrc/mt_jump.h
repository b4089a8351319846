// MT19937 jump-ahead over GF(2), host side (csrc/device_randn.hip uses the polynomials on the device).
//
// NumPy's legacy stream (np.random.randn, reference entropy/entmc_vbmc.py:64-68) is MT19937: the words x_k follow
//     x_{k+624} = x_{k+397} ^ twist(x_k, x_{k+1}),
// a linear recurrence over GF(2) on a 19937-bit state whose characteristic polynomial phi(t) is primitive, so EVERY bit
// stream (x_k[b])_k>=1 satisfies the scalar recurrence phi.  Hence for g(t) = t^n mod phi(t)
//     x_{k+n} = XOR over { i : g_i = 1 } of x_{k+i}          for every k >= 1
// (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number generators",
// 2008: the sliding-window form).  k >= 1 because only the top bit of x_0 belongs to the state: the window of a jump
// starts at x_1, and the polynomial of "the block that starts n words after x_0" is t^(n-1).
// A GPU cannot run one 12-million-word recurrence fast, but it can run 256 of 50 000 words side by side once each
// workgroup has ITS starting block -- 624 words, each the XOR of ~10 000 of the 20 560 words that follow the current
// state, the same window for every workgroup.
//
// Here: phi by Berlekamp-Massey from the recurrence itself (nothing is taken from tables), polynomial arithmetic mod
// phi, and the chain G_m = t^(m J - 1) mod phi for stream starts m J.  All of it is checked against the plain
// recurrence by tests/test_device_randn.py (CPU: vbmc_mt_jump_host).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace mtj {

constexpr int N = 624, M = 397, DEG = 19937;
constexpr int PW = (DEG + 63) / 64;  // 312 words of 64 bits hold a residue mod phi

inline uint32_t twist(uint32_t hi, uint32_t lo) {
  const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
  return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
// d = the block after s (624 words each)
inline void next_block(const uint32_t* s, uint32_t* d) {
  for (int i = 0; i < N - M; ++i) d[i] = s[i + M] ^ twist(s[i], s[i + 1]);
  for (int i = N - M; i < N - 1; ++i) d[i] = d[i - (N - M)] ^ twist(s[i], s[i + 1]);
  d[N - 1] = d[M - 1] ^ twist(s[N - 1], d[0]);
}

struct Poly {  // residue mod phi: bit i = coefficient of t^i, i < DEG
  uint64_t w[PW];
  Poly() { std::memset(w, 0, sizeof(w)); }
  bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
  void flip(int i) { w[i >> 6] ^= (uint64_t)1 << (i & 63); }
};

struct Field {
  std::vector<int> terms;  // exponents of phi's nonzero terms below DEG (phi = t^DEG + sum t^e)
  Poly tinv;               // t^-1 mod phi = (phi + 1) / t

  // Berlekamp-Massey over GF(2) on the lowest bit of x_1, x_2, ... of an arbitrary nonzero state
  void init() {
    const int L = 2 * DEG + 64;
    std::vector<uint32_t> blk(N), nxt(N);
    for (int i = 0; i < N; ++i) blk[i] = 0x9E3779B9u * (uint32_t)(i + 1) ^ 0x7F4A7C15u;  // any nonzero state
    std::vector<uint64_t> s((L + 63) / 64, 0);  // bit stream
    for (int k = 0, pos = 1; k < L; ++k) {
      if (pos == N) {
        next_block(blk.data(), nxt.data());
        blk.swap(nxt);
        pos = 0;
      }
      if (blk[pos++] & 1u) s[k >> 6] |= (uint64_t)1 << (k & 63);
    }
    // connection polynomial C (bit i = c_i, c_0 = 1): s_n = sum_{i=1..Lc} c_i s_{n-i}
    const int W = (DEG + 1 + 63) / 64 + 1;
    std::vector<uint64_t> C(W, 0), B(W, 0), T(W);
    C[0] = B[0] = 1;
    int Lc = 0, m = 1;
    // reversed stream window kept incrementally: R bit i = s_{n-i}
    std::vector<uint64_t> R(W, 0);
    for (int n = 0; n < L; ++n) {
      // R <<= 1; R[0] |= s_n
      uint64_t carry = (s[n >> 6] >> (n & 63)) & 1;
      for (int q = 0; q < W; ++q) {
        const uint64_t nc = R[q] >> 63;
        R[q] = (R[q] << 1) | carry;
        carry = nc;
      }
      // discrepancy = parity(C & R) over bits 0..Lc
      uint64_t acc = 0;
      const int qmax = (Lc >> 6) + 1;
      for (int q = 0; q < qmax && q < W; ++q) acc ^= C[q] & R[q];
      if (__builtin_parityll(acc)) {
        T = C;
        // C ^= B << m
        const int ws = m >> 6, bs = m & 63;
        for (int q = W - 1; q >= ws; --q) {
          uint64_t v = B[q - ws] << bs;
          if (bs && q - ws - 1 >= 0) v |= B[q - ws - 1] >> (64 - bs);
          C[q] ^= v;
        }
        if (2 * Lc <= n) {
          Lc = n + 1 - Lc;
          B = T;
          m = 1;
        } else {
          ++m;
        }
      } else {
        ++m;
      }
    }
    // C(t) = sum c_i t^i is the reciprocal of the characteristic polynomial: phi(t) = t^Lc C(1/t), i.e. the term t^(Lc-i)
    // for every c_i; Lc must be DEG
    terms.clear();
    if (Lc != DEG) return;  // (checked by ok())
    for (int i = 1; i <= DEG; ++i)
      if ((C[i >> 6] >> (i & 63)) & 1) terms.push_back(DEG - i);
    std::sort(terms.begin(), terms.end());
    // t^-1 = (phi + 1) / t: needs the constant term 1
    tinv = Poly();
    if (!terms.empty() && terms[0] == 0) {
      for (size_t k = 1; k < terms.size(); ++k) tinv.flip(terms[k] - 1);
      tinv.flip(DEG - 1);
    }
  }
  bool ok() const { return !terms.empty() && terms[0] == 0; }

  // r = a * b mod phi
  void mulmod(const Poly& a, const Poly& b, Poly& r) const {
    uint64_t prod[2 * PW + 1];
    std::memset(prod, 0, sizeof(prod));
    for (int i = 0; i < DEG; ++i) {
      if (!a.bit(i)) continue;
      const int ws = i >> 6, bs = i & 63;
      if (bs == 0) {
        for (int q = 0; q < PW; ++q) prod[q + ws] ^= b.w[q];
      } else {
        uint64_t carry = 0;
        for (int q = 0; q < PW; ++q) {
          prod[q + ws] ^= (b.w[q] << bs) | carry;
          carry = b.w[q] >> (64 - bs);
        }
        prod[PW + ws] ^= carry;
      }
    }
    // reduce: t^d = sum t^(d - DEG + e) for d >= DEG, from the top
    for (int d = 2 * DEG - 2; d >= DEG; --d) {
      if (!((prod[d >> 6] >> (d & 63)) & 1)) continue;
      prod[d >> 6] ^= (uint64_t)1 << (d & 63);
      const int sft = d - DEG;
      for (int e : terms) {
        const int p = sft + e;
        prod[p >> 6] ^= (uint64_t)1 << (p & 63);
      }
    }
    std::memcpy(r.w, prod, sizeof(r.w));
    r.w[PW - 1] &= ((uint64_t)1 << (DEG & 63)) - 1;
  }
  // t^e mod phi, e >= 0
  Poly tpow(uint64_t e) const {
    Poly r, base, tmp;
    r.flip(0);
    base.flip(1);
    while (e) {
      if (e & 1) {
        mulmod(r, base, tmp);
        r = tmp;
      }
      e >>= 1;
      if (e) {
        mulmod(base, base, tmp);
        base = tmp;
      }
    }
    return r;
  }
};

inline Field& field() {
  static Field f;
  static std::once_flag once;
  std::call_once(once, [] { f.init(); });
  return f;
}

// The polynomials G_m = t^(m J - 1) mod phi, m = 1 .. count, as 624 32-bit words each (bit i of the vector = coefficient
// of t^i), computed by a team of host threads: thread ranges start from t^(m0 J - 1) by square-and-multiply and go on by
// multiplying with h = t^J.
inline bool jump_polys(uint64_t J, int count, int n_threads, std::vector<uint32_t>& out) {
  Field& f = field();
  if (!f.ok()) return false;
  out.assign((size_t)count * N, 0u);
  const Poly h = f.tpow(J);
  Poly hq;  // t^(J-1)
  f.mulmod(h, f.tinv, hq);
  const int T = std::max(1, std::min(n_threads, count));
  auto work = [&](int t) {
    const int m0 = 1 + (int)((int64_t)count * t / T), m1 = 1 + (int)((int64_t)count * (t + 1) / T);
    if (m0 >= m1) return;
    Poly g, tmp;
    if (m0 == 1) {
      g = hq;
    } else {
      const Poly p = f.tpow(J * (uint64_t)(m0 - 1));
      f.mulmod(p, hq, g);
    }
    for (int m = m0; m < m1; ++m) {
      uint32_t* o = out.data() + (size_t)(m - 1) * N;
      for (int q = 0; q < PW; ++q) {
        o[2 * q] = (uint32_t)g.w[q];
        if (2 * q + 1 < N) o[2 * q + 1] = (uint32_t)(g.w[q] >> 32);
      }
      if (m + 1 < m1) {
        f.mulmod(g, h, tmp);
        g = tmp;
      }
    }
  };
  std::vector<std::thread> team;
  try {
    for (int t = 1; t < T; ++t) team.emplace_back(work, t);
  } catch (...) {
    for (auto& th : team) th.join();
    team.clear();
    for (int t = 1; t < T; ++t) work(t);
  }
  work(0);
  for (auto& th : team) th.join();
  return true;
}

// host form of the device's jump (tests): key_out = the block that starts `n_words` words after key_in[0], n_words >= 1,
// from the window x_1 .. x_(DEG + 623) generated by the plain recurrence
inline bool jump_host(const uint32_t* key_in, uint64_t n_words, uint32_t* key_out) {
  Field& f = field();
  if (!f.ok() || n_words < 1) return false;
  const Poly g = f.tpow(n_words - 1);
  const int nb = (DEG + N + N - 1) / N + 1;
  std::vector<uint32_t> win((size_t)nb * N);
  std::memcpy(win.data(), key_in, sizeof(uint32_t) * N);
  for (int b = 1; b < nb; ++b) next_block(win.data() + (size_t)(b - 1) * N, win.data() + (size_t)b * N);
  for (int j = 0; j < N; ++j) key_out[j] = 0;
  for (int i = 0; i < DEG; ++i) {
    if (!g.bit(i)) continue;
    const uint32_t* src = win.data() + 1 + i;
    for (int j = 0; j < N; ++j) key_out[j] ^= src[j];
  }
  return true;
}

}  // namespace mtj
