// wqaa_pack.hip - host-only: the CPU weight pre-processing of Matmul.transform_weight (reference:
// bitblas/ops/general_matmul/__init__.py:662-711 -> QuantCompress + LOP3Permutate, run through tvm.build(target="llvm")).
//   general_compress order  (bitblas/quantization/utils.py:54-70):  field o of a byte / word = element k % e, lowest first
//   LOP3 interleave         (bitblas/ops/lop3_permutate/lop3_permutate_impl.py:12-132): per 32-bit word, source element o
//                           -> bit lop3_dst_bit(bits, S, o) (csrc/wqaa_decode.h: the kernels' compile-time tables use the
//                           same function, so packer and decoder cannot drift apart)
// One 32-bit word at a time: the fields of 8 / 16 / 32 source bytes are gathered by SWAR shifts on 64-bit registers (plain
// order), and a layout change is four look-ups in a per-call 4 x 256 table (one per byte of the word: both orders keep fields
// on field-size boundaries, so a byte's fields move independently); rows are split over host threads for large matrices (a
// 70B checkpoint is 7e10 fields).  Pure integer work, bit exact against vectors produced by running the reference's own numpy
// functions (tests/golden/packing_golden.npz); the one-field-at-a-time loops remain for row tails and as the definition.
#include <sched.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "wqaa_common.h"
#include "wqaa_decode.h"

namespace wqaa {
namespace {

int host_threads(int64_t fields, int64_t rows) {
  if (fields < (int64_t(1) << 21) || rows < 2) return 1;      // small tensors: a thread launch costs more than the loop
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, CPU_COUNT(&set));
  n = std::min(n, 16);                                        // memory-bound beyond that; cgroup quotas are often smaller
  if (const char* e = getenv("WQAA_PACK_THREADS")) n = atoi(e);
  n = (int)std::min<int64_t>(std::max(n, 1), rows);
  return n;
}

template <typename F>
void for_rows(int64_t rows, int64_t fields_per_row, F body) {
  const int nt = host_threads(rows * fields_per_row, rows);
  if (nt <= 1) {
    body(int64_t(0), rows);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(nt - 1);
  const int64_t per = (rows + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) {
    const int64_t r0 = t * per, r1 = std::min(rows, r0 + per);
    if (r0 >= r1) continue;
    try {
      pool.emplace_back([=] { body(r0, r1); });
    } catch (...) {          // no thread to be had (pids limit of a container): this share runs here - never an exception across the C ABI
      body(r0, r1);
    }
  }
  body(int64_t(0), std::min(rows, per));
  for (auto& th : pool) th.join();
}

struct WordMap {
  int epw;       // fields per 32-bit word
  int sh[32];    // destination bit of source element o
};

bool word_map(int bits, int layout, int a_dtype, WordMap* m) {
  if (!(bits == 1 || bits == 2 || bits == 4)) return false;
  const int S = a_dtype == WQAA_I8 ? 8 : a_dtype == WQAA_I4 ? 4 : 16;
  m->epw = 32 / bits;
  for (int o = 0; o < m->epw; ++o) m->sh[o] = layout == WQAA_LAYOUT_LOP3 ? lop3_dst_bit(bits, S, o) : o * bits;
  return true;
}

// ---- SWAR: EPW source bytes (one field each, low BITS bits) <-> one 32-bit word in plain order (field o at bit o * BITS) ----
inline uint64_t load8(const uint8_t* p) {
  uint64_t x;
  memcpy(&x, p, 8);
  return x;
}
template <int BITS>
inline uint32_t gather_word(const uint8_t* src) {
  if constexpr (BITS == 4) {
    uint64_t x = load8(src) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    return (uint32_t)(x | (x >> 16));
  } else if constexpr (BITS == 2) {
    uint32_t w = 0;
    for (int h = 0; h < 2; ++h) {
      uint64_t x = load8(src + 8 * h) & 0x0303030303030303ull;
      x = (x | (x >> 6)) & 0x000F000F000F000Full;
      x = (x | (x >> 12)) & 0x000000FF000000FFull;
      w |= (uint32_t)((x | (x >> 24)) & 0xFFFFu) << (16 * h);
    }
    return w;
  } else {
    uint32_t w = 0;
    for (int q = 0; q < 4; ++q) {
      // bit 0 of byte i -> bit 56 + i: the partial products of 2^(56 - 7 j) land on distinct bits or beyond bit 63
      const uint64_t x = load8(src + 8 * q) & 0x0101010101010101ull;
      w |= (uint32_t)((x * 0x0102040810204080ull) >> 56) << (8 * q);
    }
    return w;
  }
}
template <int BITS>
inline void scatter_word(uint32_t w, uint8_t* dst) {
  if constexpr (BITS == 4) {
    uint64_t x = w;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    memcpy(dst, &x, 8);
  } else if constexpr (BITS == 2) {
    for (int h = 0; h < 2; ++h) {
      uint64_t x = (w >> (16 * h)) & 0xFFFFu;
      x = (x | (x << 24)) & 0x000000FF000000FFull;
      x = (x | (x << 12)) & 0x000F000F000F000Full;
      x = (x | (x << 6)) & 0x0303030303030303ull;
      memcpy(dst + 8 * h, &x, 8);
    }
  } else {
    for (int q = 0; q < 4; ++q) {
      uint64_t x = ((uint64_t)((w >> (8 * q)) & 0xFFu) * 0x0101010101010101ull) & 0x8040201008040201ull;   // byte i keeps bit i
      x = ((x + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;                                        // ... as 0 / 1
      memcpy(dst + 8 * q, &x, 8);
    }
  }
}

// layout change word -> word as four byte look-ups: t[j][b] = where the fields of byte j (value b) of the source word go
struct ByteLut {
  uint32_t t[4][256];
  uint32_t apply(uint32_t w) const { return t[0][w & 0xFF] | t[1][(w >> 8) & 0xFF] | t[2][(w >> 16) & 0xFF] | t[3][w >> 24]; }
};
// from.sh[o] = bit of element o in the source word, to.sh[o] = its bit in the destination word
void make_lut(int bits, const WordMap& from, const WordMap& to, ByteLut* lut) {
  const int epb = 8 / bits;
  const uint32_t mask = (1u << bits) - 1u;
  int elem_at[32];                                   // element whose field sits at field position p of the source word
  for (int o = 0; o < from.epw; ++o) elem_at[from.sh[o] / bits] = o;
  for (int j = 0; j < 4; ++j)
    for (int b = 0; b < 256; ++b) {
      uint32_t v = 0;
      for (int k = 0; k < epb; ++k) v |= (((uint32_t)b >> (k * bits)) & mask) << to.sh[elem_at[j * epb + k]];
      lut->t[j][b] = v;
    }
}

// BITS is a template parameter so the 8 / 16 / 32-field loops unroll with constant trip counts
template <int BITS>
void pack_rows(const int8_t* codes, int64_t r0, int64_t r1, int64_t cols, const ByteLut* lut, bool words, uint8_t* out) {
  constexpr int EPW = 32 / BITS, EPB = 8 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  const int64_t row_bytes = cols / EPB;
  for (int64_t r = r0; r < r1; ++r) {
    const uint8_t* src = reinterpret_cast<const uint8_t*>(codes) + r * cols;
    uint8_t* dst = out + r * row_bytes;
    int64_t c = 0;
    if (words) {
      for (; c + EPW <= cols; c += EPW) {
        uint32_t w = gather_word<BITS>(src + c);
        if (lut) w = lut->apply(w);
        memcpy(dst + (c / EPW) * 4, &w, 4);
      }
    }
    for (; c < cols; c += EPB) {      // plain tail of a row that is not a whole number of words, byte at a time
      uint8_t b = 0;
      for (int k = 0; k < EPB; ++k) b |= (uint8_t)((src[c + k] & MASK) << (BITS * k));
      dst[c / EPB] = b;
    }
  }
}

template <int BITS>
void unpack_rows(const uint8_t* packed, int64_t r0, int64_t r1, int64_t cols, const ByteLut* lut, bool words, int8_t* codes) {
  constexpr int EPW = 32 / BITS, EPB = 8 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  const int64_t row_bytes = cols / EPB;
  for (int64_t r = r0; r < r1; ++r) {
    const uint8_t* src = packed + r * row_bytes;
    int8_t* dst = codes + r * cols;
    int64_t c = 0;
    if (words) {
      for (; c + EPW <= cols; c += EPW) {
        uint32_t w;
        memcpy(&w, src + (c / EPW) * 4, 4);
        if (lut) w = lut->apply(w);                   // back to plain order
        scatter_word<BITS>(w, reinterpret_cast<uint8_t*>(dst) + c);
      }
    }
    for (; c < cols; ++c) dst[c] = (int8_t)((src[c / EPB] >> (BITS * (c % EPB))) & MASK);
  }
}

// word -> word: from one layout's field order to the other's, no byte-per-field intermediate
void remap_rows(const uint8_t* in, int64_t r0, int64_t r1, int64_t row_words, const ByteLut& lut, uint8_t* out) {
  for (int64_t r = r0; r < r1; ++r) {
    const uint8_t* src = in + r * row_words * 4;
    uint8_t* dst = out + r * row_words * 4;
    for (int64_t i = 0; i < row_words; ++i) {
      uint32_t w;
      memcpy(&w, src + i * 4, 4);
      w = lut.apply(w);
      memcpy(dst + i * 4, &w, 4);
    }
  }
}

bool check_common(const char* what, const void* a, const void* b, int64_t rows, int64_t cols, int bits) {
  if (!a || !b || rows < 0 || cols < 0 || !(bits == 1 || bits == 2 || bits == 4 || bits == 8)) {
    set_error(WQAA_ERR_BAD_DESC, "%s: bad arguments (bits=%d)", what, bits);
    return false;
  }
  if (cols % (8 / bits)) {
    set_error(WQAA_ERR_BAD_DESC, "%s: cols=%ld not a multiple of %d", what, (long)cols, 8 / bits);
    return false;
  }
  return true;
}

}  // namespace
}  // namespace wqaa

using namespace wqaa;

extern "C" {

int wqaa_pack_weight(const int8_t* codes, int64_t rows, int64_t cols, int bits, int layout, int a_dtype, int8_t* out) {
  if (!check_common("pack_weight", codes, out, rows, cols, bits)) return WQAA_ERR_BAD_DESC;
  if (bits == 8) {
    memcpy(out, codes, (size_t)(rows * cols));
    return WQAA_OK;
  }
  const int64_t row_bytes = cols * bits / 8;
  if (layout == WQAA_LAYOUT_LOP3 && row_bytes % 4) {
    set_error(WQAA_ERR_BAD_DESC, "pack_weight: LOP3 layout needs K*bits %% 32 == 0");
    return WQAA_ERR_BAD_DESC;
  }
  WordMap m, plain;
  word_map(bits, layout, a_dtype, &m);
  word_map(bits, WQAA_LAYOUT_PLAIN, a_dtype, &plain);
  ByteLut lut_store;
  const ByteLut* lut = nullptr;
  if (layout == WQAA_LAYOUT_LOP3) {
    make_lut(bits, plain, m, &lut_store);
    lut = &lut_store;
  }
  const bool words = layout == WQAA_LAYOUT_LOP3 || row_bytes % 4 == 0;
  uint8_t* o = reinterpret_cast<uint8_t*>(out);
  for_rows(rows, cols, [=](int64_t r0, int64_t r1) {
    if (bits == 4) pack_rows<4>(codes, r0, r1, cols, lut, words, o);
    else if (bits == 2) pack_rows<2>(codes, r0, r1, cols, lut, words, o);
    else pack_rows<1>(codes, r0, r1, cols, lut, words, o);
  });
  return WQAA_OK;
}

int wqaa_unpack_weight(const int8_t* packed, int64_t rows, int64_t cols, int bits, int layout, int a_dtype, int8_t* codes) {
  if (!check_common("unpack_weight", packed, codes, rows, cols, bits)) return WQAA_ERR_BAD_DESC;
  if (bits == 8) {
    memcpy(codes, packed, (size_t)(rows * cols));
    return WQAA_OK;
  }
  const int64_t row_bytes = cols * bits / 8;
  if (layout == WQAA_LAYOUT_LOP3 && row_bytes % 4) {
    set_error(WQAA_ERR_BAD_DESC, "unpack_weight: LOP3 layout needs K*bits %% 32 == 0");
    return WQAA_ERR_BAD_DESC;
  }
  WordMap m, plain;
  word_map(bits, layout, a_dtype, &m);
  word_map(bits, WQAA_LAYOUT_PLAIN, a_dtype, &plain);
  ByteLut lut_store;
  const ByteLut* lut = nullptr;
  if (layout == WQAA_LAYOUT_LOP3) {
    make_lut(bits, m, plain, &lut_store);
    lut = &lut_store;
  }
  const bool words = layout == WQAA_LAYOUT_LOP3 || row_bytes % 4 == 0;
  const uint8_t* p = reinterpret_cast<const uint8_t*>(packed);
  for_rows(rows, cols, [=](int64_t r0, int64_t r1) {
    if (bits == 4) unpack_rows<4>(p, r0, r1, cols, lut, words, codes);
    else if (bits == 2) unpack_rows<2>(p, r0, r1, cols, lut, words, codes);
    else unpack_rows<1>(p, r0, r1, cols, lut, words, codes);
  });
  return WQAA_OK;
}

int wqaa_relayout_weight(const int8_t* packed, int64_t rows, int64_t row_bytes, int bits, int from_layout, int to_layout, int a_dtype,
                         int8_t* out) {
  if (!packed || !out || rows < 0 || row_bytes < 0 || !(bits == 1 || bits == 2 || bits == 4 || bits == 8)) {
    set_error(WQAA_ERR_BAD_DESC, "relayout_weight: bad arguments (bits=%d)", bits);
    return WQAA_ERR_BAD_DESC;
  }
  if (bits == 8 || from_layout == to_layout) {
    if (out != packed) memcpy(out, packed, (size_t)(rows * row_bytes));
    return WQAA_OK;
  }
  if (row_bytes % 4) {
    set_error(WQAA_ERR_BAD_DESC, "relayout_weight: LOP3 layout needs K*bits %% 32 == 0");
    return WQAA_ERR_BAD_DESC;
  }
  WordMap from, to;
  word_map(bits, from_layout, a_dtype, &from);
  word_map(bits, to_layout, a_dtype, &to);
  const uint8_t* p = reinterpret_cast<const uint8_t*>(packed);
  uint8_t* o = reinterpret_cast<uint8_t*>(out);
  const int64_t row_words = row_bytes / 4;
  ByteLut lut;
  make_lut(bits, from, to, &lut);
  for_rows(rows, row_bytes * 8 / bits, [=, &lut](int64_t r0, int64_t r1) { remap_rows(p, r0, r1, row_words, lut, o); });
  return WQAA_OK;
}

}  // extern "C"
