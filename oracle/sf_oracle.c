/*
 * sf_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see sf_oracle.h).
 *
 * Part 1 restates /root/reference/src/decompress.cpp (file:line cited per
 * function).  Part 2 is the scalar specification of the block-parallel DEFLATE
 * encoder implemented by starflate_amd/csrc/ (no reference counterpart).
 */
#include "sf_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ====================================================================== */
/* Part 1: decoder restatement                                            */
/* ====================================================================== */

/* huffman/src/bit_span.hpp:18-183: LSB-first bit view; stream bit i is
 * (byte[i/8] >> (i%8)) & 1  (bit_span.hpp:46-53). */
typedef struct {
  const uint8_t* p;
  size_t nbits; /* total bits in the span */
  size_t pos;   /* bits consumed */
} bitr;

static inline unsigned br_bit(const bitr* b, size_t i) { return (b->p[i >> 3] >> (i & 7)) & 1u; }
static inline size_t br_left(const bitr* b) { return b->nbits - b->pos; }

/* src/decompress.cpp:94-114 pop_bits<T>: n bits, LSB-first.  The reference
 * asserts "bits contains at least n bits"; here that case returns -1. */
static int pop_bits(bitr* b, unsigned n, unsigned* out) {
  if (br_left(b) < n) return -1;
  unsigned r = 0;
  for (unsigned i = 0; i < n; i++) r |= br_bit(b, b->pos + i) << i;
  b->pos += n;
  *out = r;
  return 0;
}

/* A canonical table in the shape huffman::table has after canonicalize()
 * (huffman/src/table.hpp:177-216): entries sorted by (bitsize, symbol), codes
 * consecutive within a bitsize, `base_code <<= delta` on a bitsize change.
 * table::find (table.hpp:426-452) walks bitsize groups using `skip` = group size;
 * first[len] / count[len] / index[len] are that walk, tabulated. */
#define MAXLEN 32
typedef struct {
  uint32_t count[MAXLEN + 1];
  uint64_t first[MAXLEN + 1];
  uint32_t index[MAXLEN + 1];
  uint16_t syms[320];
  uint64_t code[320]; /* by sorted index */
  uint32_t n;         /* entries */
  uint32_t maxlen;
} ctab;

/* Returns 1 when the lengths claim more code space than there is (Kraft sum > 1): canonicalize() then makes a
 * code whose value does not fit its bitsize, which the reference's code constructor asserts against
 * (huffman/src/code.hpp:33-44, "`value` exceeds `bitsize`") -- SFO_ERROR by this file's rule for asserts. */
static int ctab_build(ctab* t, const uint8_t* bitsize, uint32_t nsyms) {
  memset(t, 0, sizeof *t);
  for (uint32_t s = 0; s < nsyms; s++)
    if (bitsize[s]) {
      t->count[bitsize[s]]++;
      if (bitsize[s] > t->maxlen) t->maxlen = bitsize[s];
    }
  uint32_t idx = 0;
  for (uint32_t l = 1; l <= MAXLEN; l++) {
    t->index[l] = idx;
    idx += t->count[l];
  }
  t->n = idx;
  uint32_t fill[MAXLEN + 1];
  memcpy(fill, t->index, sizeof fill);
  for (uint32_t s = 0; s < nsyms; s++)
    if (bitsize[s]) t->syms[fill[bitsize[s]]++] = (uint16_t)s;
  /* canonicalize(), table.hpp:190-210 */
  uint64_t base_code = 0;
  uint32_t cur_len = 0;
  int over = 0;
  for (uint32_t l = 1; l <= MAXLEN; l++) {
    for (uint32_t k = 0; k < t->count[l]; k++) {
      uint64_t value;
      if (cur_len == l) {
        value = base_code; /* next_code.value()+1 == base_code */
      } else {
        base_code <<= (l - cur_len);
        value = base_code;
        cur_len = l;
      }
      if (k == 0) t->first[l] = value;
      t->code[t->index[l] + k] = value;
      if (value >> l) over = 1;
      ++base_code;
    }
  }
  return over;
}

/* huffman/src/decode.hpp:83-102 decode_one: bits enter the code MSB-first
 * (`current_code << bit`, code.hpp:90-96).  Returns encoded size, 0 = invalid
 * (decode_result::kInvalidEncodedSize). */
static unsigned decode_one(const ctab* t, const bitr* b, unsigned* sym) {
  uint64_t code = 0;
  size_t left = br_left(b);
  for (unsigned len = 1;; len++) {
    if (len > left) return 0; /* `for (auto bit : bits)` ran out */
    code = (code << 1) | br_bit(b, b->pos + len - 1);
    if (len <= MAXLEN && t->count[len]) {
      uint64_t d = code - t->first[len]; /* unsigned, as table.hpp:441 */
      if (d < t->count[len]) {
        *sym = t->syms[t->index[len] + d];
        return len;
      }
    }
    if (len >= t->maxlen) return 0; /* find() returned end() (decode.hpp:96-98) */
  }
}

void sfo_canonical_codes(const uint8_t* bitsize, uint32_t n, uint32_t* code_out) {
  ctab t;
  ctab_build(&t, bitsize, n);
  for (uint32_t s = 0; s < n; s++) code_out[s] = 0;
  for (uint32_t k = 0; k < t.n; k++) code_out[t.syms[k]] = (uint32_t)t.code[k];
}

size_t sfo_huffman_decode(const uint8_t* bitsize, uint32_t nsyms, const uint8_t* src,
                          size_t nbits, uint16_t* out, size_t out_cap) {
  ctab t;
  ctab_build(&t, bitsize, nsyms);
  bitr b = {src, nbits, 0};
  size_t n = 0;
  while (br_left(&b) > 0 && n < out_cap) { /* decode.hpp:29-36 */
    unsigned sym;
    unsigned sz = decode_one(&t, &b, &sym);
    if (!sz) break;
    out[n++] = (uint16_t)sym;
    b.pos += sz;
  }
  return n;
}

/* src/decompress.cpp:53-84 */
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                      2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                      31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                       6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint16_t dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                       33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                       1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
/* src/decompress.cpp:250-251 */
static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* src/decompress.cpp:370-385 */
static int read_header(bitr* b, int* final, int* type) {
  if (br_left(b) < 3) return SFO_INVALID_BLOCK_HEADER;
  int t = (int)(br_bit(b, b->pos + 1) | (br_bit(b, b->pos + 2) << 1));
  if (t == 3) return SFO_INVALID_BLOCK_HEADER;
  *final = (int)br_bit(b, b->pos);
  *type = t;
  b->pos += 3;
  return SFO_SUCCESS;
}

int sfo_read_header(const uint8_t* src, size_t src_bits, int* final, int* type) {
  bitr b = {src, src_bits, 0};
  return read_header(&b, final, type);
}

/* src/decompress.cpp:388-398 */
void sfo_copy_from_before(uint16_t distance, uint8_t* dst, uint16_t n) {
  ptrdiff_t left = n;
  const uint8_t* src = dst - distance;
  while (left > 0) {
    ptrdiff_t k = left < (dst - src) ? left : (dst - src);
    memcpy(dst, src, (size_t)k);
    dst += k;
    left -= k;
  }
}

/* src/decompress.cpp:197-242 with :122-187 inlined */
static int block_huffman(bitr* b, uint8_t* dst, size_t dst_cap, size_t* written, const ctab* lt,
                         const ctab* dt) {
  for (;;) {
    unsigned sym;
    unsigned sz = decode_one(lt, b, &sym);
    if (!sz) return SFO_INVALID_LIT_OR_LEN; /* :214-216 */
    b->pos += sz;
    if (sym < 256) { /* :125-127, :146-155 */
      if (dst_cap - *written < 1) return SFO_DST_TOO_SMALL;
      dst[(*written)++] = (uint8_t)sym;
      continue;
    }
    if (sym == 256) return SFO_SUCCESS; /* :128-130, :221-223 */
    if (sym > 285) return SFO_INVALID_LIT_OR_LEN; /* :131-133 */
    unsigned len;
    if (sym == 285) {
      len = 258; /* :134-136 */
    } else {
      unsigned ex;
      if (pop_bits(b, len_extra[sym - 257], &ex)) return SFO_ERROR;
      len = len_base[sym - 257] + ex;
    }
    unsigned dsym;
    sz = decode_one(dt, b, &dsym);
    if (!sz) return SFO_INVALID_DISTANCE; /* :166-168 */
    b->pos += sz;
    if (dsym >= 30) return SFO_INVALID_LIT_OR_LEN; /* :170-172 */
    unsigned ex;
    if (pop_bits(b, dist_extra[dsym], &ex)) return SFO_ERROR;
    unsigned distance = dist_base[dsym] + ex;
    if (distance > *written) return SFO_INVALID_DISTANCE; /* :177-179 */
    if (dst_cap - *written < len) return SFO_DST_TOO_SMALL; /* :180-182 */
    sfo_copy_from_before((uint16_t)distance, dst + *written, (uint16_t)len);
    *written += len;
  }
}

/* src/decompress.cpp:253-312: one code-length sequence, own zeroed vector, own
 * RLE state.  The reference does not bound-check runs (hazards A/B in
 * SURVEY.md section 0); those inputs return SFO_ERROR here. */
static int read_code_lengths(bitr* b, const ctab* clt, unsigned n_codes, uint8_t* out) {
  memset(out, 0, n_codes);
  for (unsigned i = 0; i < n_codes; i++) {
    unsigned sym;
    unsigned sz = decode_one(clt, b, &sym);
    if (!sz) return SFO_INVALID_LIT_OR_LEN; /* :263-265 */
    b->pos += sz;
    if (sym < 16) {
      out[i] = (uint8_t)sym;
    } else {
      unsigned nb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
      unsigned base = sym == 18 ? 11 : 3;
      unsigned rep;
      if (sym > 18) return SFO_INVALID_LIT_OR_LEN; /* :297-299 */
      if (pop_bits(b, nb, &rep)) return SFO_ERROR;
      rep += base;
      if (sym == 16 && i == 0) return SFO_ERROR;  /* reads code_bitsizes[-1] in the reference */
      if (i + rep > n_codes) return SFO_ERROR;    /* writes past the vector in the reference */
      uint8_t v = sym == 16 ? out[i - 1] : 0;
      for (unsigned j = 0; j < rep; j++) out[i + j] = v;
      i += rep - 1;
    }
  }
  return SFO_SUCCESS;
}

/* src/decompress.cpp:314-367 */
static int read_dynamic_tables(bitr* b, ctab* lt, ctab* dt) {
  unsigned hlit, hdist, hclen;
  if (pop_bits(b, 5, &hlit) || pop_bits(b, 5, &hdist) || pop_bits(b, 4, &hclen)) return SFO_ERROR;
  unsigned n_len = 257 + hlit, n_dist = 1 + hdist, n_cl = 4 + hclen;
  uint8_t cl_bits[19] = {0};
  for (unsigned i = 0; i < n_cl; i++) {
    unsigned v;
    if (pop_bits(b, 3, &v)) return SFO_ERROR;
    cl_bits[cl_order[i]] = (uint8_t)v;
  }
  ctab clt;
  if (ctab_build(&clt, cl_bits, 19)) return SFO_ERROR;
  uint8_t lens[320];
  int st = read_code_lengths(b, &clt, n_len, lens);
  if (st) return st;
  if (ctab_build(lt, lens, n_len)) return SFO_ERROR;
  st = read_code_lengths(b, &clt, n_dist, lens);
  if (st) return st;
  if (ctab_build(dt, lens, n_dist)) return SFO_ERROR;
  return SFO_SUCCESS;
}

static ctab g_fixed_l, g_fixed_d;
static int g_fixed_ready;
/* src/decompress.cpp:16-40: 288 lit/len symbols (286/287 exist in the table and
 * are rejected after decode), 32 distance symbols of 5 bits. */
static void fixed_tables(void) {
  if (g_fixed_ready) return;
  uint8_t l[288], d[32];
  for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
  for (int i = 0; i < 32; i++) d[i] = 5;
  ctab_build(&g_fixed_l, l, 288);
  ctab_build(&g_fixed_d, d, 32);
  g_fixed_ready = 1;
}

/* src/decompress.cpp:402-461 */
int sfo_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap,
                   size_t* dst_written) {
  fixed_tables();
  bitr b = {src, src_len * 8, 0};
  size_t written = 0;
  int st = SFO_SUCCESS;
  for (int was_final = 0; !was_final;) {
    int final, type;
    st = read_header(&b, &final, &type);
    if (st) break;
    was_final = final;
    if (type == 0) {                 /* :416-436 */
      b.pos = (b.pos + 7) & ~(size_t)7; /* consume_to_byte_boundary */
      if (br_left(&b) < 32) { st = SFO_ERROR; break; } /* pop_16 asserts in the reference */
      size_t byte = b.pos >> 3;
      unsigned len = src[byte] | (src[byte + 1] << 8);
      unsigned nlen = src[byte + 2] | (src[byte + 3] << 8);
      b.pos += 32;
      if (len != (uint16_t)~nlen) { st = SFO_NO_COMPRESSION_LEN_MISMATCH; break; }
      if (br_left(&b) < (size_t)len * 8) { st = SFO_SRC_TOO_SMALL; break; }
      if (dst_cap - written < len) { st = SFO_DST_TOO_SMALL; break; }
      memcpy(dst + written, src + (b.pos >> 3), len);
      b.pos += (size_t)len * 8;
      written += len;
    } else if (type == 1) { /* :437-446 */
      st = block_huffman(&b, dst, dst_cap, &written, &g_fixed_l, &g_fixed_d);
      if (st) break;
    } else { /* :447-458 */
      ctab lt, dt;
      st = read_dynamic_tables(&b, &lt, &dt);
      if (st) break;
      st = block_huffman(&b, dst, dst_cap, &written, &lt, &dt);
      if (st) break;
    }
  }
  if (dst_written) *dst_written = written;
  return st;
}

/* ====================================================================== */
/* Part 2: encoder specification                                          */
/* ====================================================================== */

void sfo_default_params(sfo_params* p) {
  memset(p, 0, sizeof *p);
  p->chunk_bytes = 32768;
  p->step = 1024;
  p->hash_bits = 13;
  p->region_bytes = 512;
  p->min_match = 4;
  p->lazy = 3;
  p->final_stream = 1;
  p->strategy = 0;
  p->depth = 2;
  p->use_near = 1;
  p->long_hash_bytes = 0;
  p->chain_depth = 0;
  p->cap = 16;
  p->fast_skip = 1;
  p->far4_dist = 4096;
  p->strip_bytes = 0;
  p->rank_bytes = 8;
  p->run_dist1 = 1;
  p->stride2 = 1;
}

static inline uint32_t load32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static inline uint32_t hash_of(uint32_t v, const sfo_params* p) {
  if (p->min_match == 3) v &= 0xFFFFFFu;
  return (v * 2654435761u) >> (32 - p->hash_bits);
}

static uint32_t match_len(const uint8_t* d, uint32_t i, uint32_t c, uint32_t maxlen) {
  uint32_t l = 0;
  while (l < maxlen && d[i + l] == d[c + l]) l++;
  return l;
}

#define NONE 0xFFFFFFFFu

/* hash of long_hash_bytes (5..8) bytes: two 32-bit multiplicative hashes, of the first four bytes and of the rest, xored */
static inline uint32_t hash_long(const uint8_t* q, const sfo_params* p) {
  uint32_t lo = load32(q), hi = 0;
  for (uint32_t k = 4; k < p->long_hash_bytes; k++) hi |= (uint32_t)q[k] << (8 * (k - 4));
  return ((lo * 2654435761u) ^ (hi * 0x85EBCA6Bu)) >> (32 - p->hash_bits);
}

/* step-table entry <-> position: ((step+1) << 12) | (4095 - t), t = pos - step*W */
static inline uint32_t ent_pos(uint32_t v, uint32_t W) { return ((v >> 12) - 1) * W + (4095 - (v & 4095)); }

/*
 * Stage n1 over one STRIP: `strip_bytes` of input coded independently of everything before it
 * (the unit a workgroup owns; the decoder only requires distance <= bytes already written,
 * /root/reference/src/decompress.cpp:178, so matches may reach back across the strip's DEFLATE
 * blocks).  A match reaches back at most SFO_WINDOW bytes and never before the strip's start.
 *
 * Positions are hashed in steps of p->step consecutive positions, numbered from the strip's start.
 * Every position of a step first reads its candidates, then the whole step is
 * inserted -- exactly what a workgroup does between two barriers, and
 * independent of the order in which threads run.
 *
 * Short table S (hash of min_match bytes) and optional long table L (hash of
 * long_hash_bytes bytes), each with `depth` history levels:
 *   level 0: u32 entry = ((step+1) << 12) | (4095 - t), updated by MAX, so it
 *            names the FIRST position of the LATEST step that contained the hash;
 *   level k>0: the value level k-1 held before the most recent step that
 *            inserted this hash (every inserting position of a step writes the
 *            same value, so the write is order-independent).
 * Candidates of position i: per table, the levels as read before this step's
 * insertions ("far", all < step start) and, with use_near, level 0 after the
 * insertions if it names a position < i ("near": first same-hash position of
 * this step).  Candidates farther back than SFO_WINDOW are ignored.
 * Best = longest; ties -> smallest distance.
 * chain_depth > 0 (SFH_EFFORT_BEST / _ULTRA on the GPU): exact hash chains, zlib's structure -- H[hash] is the latest
 * position with that hash, prev[i] the one before i; EVERY position is inserted and searched, most recent candidate
 * first, at most chain_depth of them and none farther back than SFO_WINDOW.  Every candidate is compared up to `cap`
 * bytes (0: up to the longest match); the longest wins, the first found (the nearest) on ties -- so the walk may stop
 * at the first candidate that reaches `cap`, which changes nothing.  far4_dist applies; depth / use_near / stride2 /
 * long_hash_bytes / rank_bytes do not.  A capped match is extended by the parse, like any other.
 */
typedef struct {
  const sfo_params* p;
  uint8_t* d;        /* strip bytes + 16 bytes of zero padding */
  uint32_t n;        /* strip length */
  uint32_t* T;       /* [table][level][hash] */
  uint32_t* far;     /* [step position][table][level] */
  uint32_t* H;       /* chain_depth analysis mode: heads / links */
  uint32_t* prev;
  uint16_t* len16;   /* [n] best match per position (0: none) */
  uint16_t* dist16;
} matcher;

static int matcher_init(matcher* m, const uint8_t* src, uint32_t n, const sfo_params* p) {
  const uint32_t HS = 1u << p->hash_bits, D = p->depth ? p->depth : 1, NT = p->long_hash_bytes ? 2 : 1;
  memset(m, 0, sizeof *m);
  m->p = p;
  m->n = n;
  m->d = (uint8_t*)calloc((size_t)n + 16, 1);
  m->len16 = (uint16_t*)calloc((size_t)n + 1, 2);
  m->dist16 = (uint16_t*)calloc((size_t)n + 1, 2);
  if (p->chain_depth || p->recent) {
    m->H = (uint32_t*)malloc((size_t)HS * 4);
    m->prev = (uint32_t*)malloc((size_t)(n + 1) * 4);
    if (m->H) for (uint32_t k = 0; k < HS; k++) m->H[k] = 0xFFFFFFFFu;
    if (p->recent && !p->chain_depth) { /* hi half of the buckets; the buckets as read before a step */
      m->T = (uint32_t*)malloc((size_t)HS * 4);
      m->far = (uint32_t*)malloc((size_t)p->step * 2 * 4);
      if (m->T) for (uint32_t k = 0; k < HS; k++) m->T[k] = 0xFFFFFFFFu;
      if (!m->T || !m->far) return -1;
    }
  } else {
    m->T = (uint32_t*)calloc((size_t)NT * D * HS, 4);
    m->far = (uint32_t*)malloc((size_t)p->step * NT * D * 4);
  }
  if (!m->d || !m->len16 || !m->dist16 || ((p->chain_depth || p->recent) ? (!m->H || !m->prev) : (!m->T || !m->far))) return -1;
  if (n) memcpy(m->d, src, n);
  return 0;
}
static void matcher_free(matcher* m) {
  free(m->d); free(m->len16); free(m->dist16); free(m->T); free(m->far); free(m->H); free(m->prev);
}

/* steps [s0, s1) of the strip: candidates, insertion, best match per position */
static void match_steps(matcher* m, uint32_t s0, uint32_t s1) {
  const sfo_params* p = m->p;
  const uint8_t* d = m->d;
  const uint32_t n = m->n;
  const uint32_t W = p->step, HS = 1u << p->hash_bits, R = p->region_bytes, MM = p->min_match;
  const uint32_t D = p->depth ? p->depth : 1, LB = p->long_hash_bytes;
  const uint32_t NT = LB ? 2 : 1;
  uint16_t *len16 = m->len16, *dist16 = m->dist16;

  if (p->chain_depth) {
    uint32_t *H = m->H, *prev = m->prev;
    uint32_t b = s0 * W, e = (uint64_t)s1 * W < n ? s1 * W : n;
    for (uint32_t i = b; i < e && i + MM <= n; i++) {
      uint32_t h = hash_of(load32(d + i), p);
      uint32_t rend = (i / R + 1) * R;
      uint32_t maxlen = n - i < 258 ? n - i : 258;
      if (rend - i < maxlen) maxlen = rend - i;
      uint32_t cmplen = (p->cap && p->cap < maxlen) ? p->cap : maxlen;
      uint32_t best = 0, bc = 0, c = H[h];
      if (p->x_long_levels & 64) { /* analysis: candidates ranked by TAGS (hashes of bytes 4..7, 8..11, 12..15), the best one verified */
        const uint32_t TB = (p->x_long_levels >> 8) ? (p->x_long_levels >> 8) : 5;
        uint32_t lvl_best = 0, have = 0;
        uint32_t ti[3];
        for (uint32_t q = 0; q < 3; q++) ti[q] = (load32(d + i + 4 + 4 * q) * 2654435761u) >> (32 - TB);
        for (uint32_t k = 0; k < p->chain_depth && c != NONE && i - c <= (p->x_window ? p->x_window : SFO_WINDOW); k++, c = prev[c]) {
          uint32_t lvl = 1;
          for (uint32_t q = 0; q < 3; q++) {
            if (((load32(d + c + 4 + 4 * q) * 2654435761u) >> (32 - TB)) != ti[q]) break;
            lvl++;
          }
          if (!have || lvl > lvl_best) { lvl_best = lvl; bc = c; have = 1; }
          if (lvl_best == 4) break;
        }
        if (have) best = match_len(d, i, bc, cmplen);
      } else
      for (uint32_t k = 0; k < p->chain_depth && c != NONE && i - c <= (p->x_window ? p->x_window : SFO_WINDOW); k++, c = prev[c]) {
        uint32_t l = match_len(d, i, c, cmplen);
        if (l > best) { best = l; bc = c; }
        if (p->cap && best == p->cap) break; /* nothing later can be longer at match time, and the first found wins ties */
      }
      if (p->far4_dist && best == 4 && i - bc > p->far4_dist) best = 0;
      if (best >= MM) { len16[i] = (uint16_t)best; dist16[i] = (uint16_t)(i - bc); }
      prev[i] = H[h];
      H[h] = i;
    }
    return;
  }

  if (p->recent) {
    /* EXACT RECENCY step tables.  lo[h]: the latest position with hash h; hi[h]: what lo[h] held before the most recent
     * step that inserted h (every inserting position of a step writes the same value); prev[i]: what lo named just
     * before i went in, i.e. the nearest earlier position with i's hash.  A position's candidates:
     *   the link chain  c1 = prev[i], c2 = prev[c1], ... : at most near_depth of them; a link is only followed FROM a
     *                   position of the current step or the link_steps - 1 steps before it (the links kept);
     *   lo and hi as they stood before the step (lo == c1 for the step's first position with the hash).
     * Ranking, winner, far4 and the odd positions' inheritance are the step tables'. */
    uint32_t *lo = m->H, *hi = m->T, *prev = m->prev, *far = m->far;
    const uint32_t ND = p->near_depth ? p->near_depth : 1, LS = p->link_steps ? p->link_steps : 1;
    const uint32_t WIN = p->x_window ? p->x_window : SFO_WINDOW;
    for (uint32_t s = s0; s < s1; s++) {
      uint32_t b = s * W;
      if (b >= n) break;
      uint32_t e = b + W < n ? b + W : n;
      for (uint32_t i = b; i < e; i++) {
        far[2 * (i - b)] = far[2 * (i - b) + 1] = NONE;
        if (i + MM > n) continue;
        uint32_t h = hash_of(load32(d + i), p);
        far[2 * (i - b)] = lo[h];
        far[2 * (i - b) + 1] = hi[h];
      }
      for (uint32_t i = b; i < e; i++) {
        if (i + MM > n) continue;
        uint32_t h = hash_of(load32(d + i), p);
        prev[i] = lo[h];
        lo[h] = i;
        hi[h] = far[2 * (i - b)];
      }
      const uint32_t ring_lo = s + 1 >= LS ? (s + 1 - LS) * W : 0;
      for (uint32_t i = b; i < e; i++) {
        if (p->stride2 && (i & 1)) continue;
        if (i + MM > n) continue;
        uint32_t rend = (i / R + 1) * R;
        uint32_t maxlen = n - i < 258 ? n - i : 258;
        if (rend - i < maxlen) maxlen = rend - i;
        uint32_t cmplen = (p->cap && p->cap < maxlen) ? p->cap : maxlen;
        uint32_t cand[40], nc = 0;
        for (uint32_t k = 0, c = prev[i]; k < ND && k < 32 && c != NONE; k++) {
          cand[nc++] = c;
          if (c < ring_lo) break;
          c = prev[c];
        }
        if (far[2 * (i - b)] != NONE) cand[nc++] = far[2 * (i - b)];
        if (far[2 * (i - b) + 1] != NONE && !(p->x_long_levels & 16)) cand[nc++] = far[2 * (i - b) + 1]; /* analysis: 16 = no hi level */
        uint32_t best = 0, bdist = 0;
        for (uint32_t k = 0; k < nc; k++) {
          uint32_t dist = i - cand[k];
          if (dist > WIN) continue;
          uint32_t l = match_len(d, i, cand[k], cmplen);
          if (p->rank_bytes && l > p->rank_bytes) l = p->rank_bytes;
          if (l > best || (l == best && l && dist < bdist)) { best = l; bdist = dist; }
        }
        if (p->rank_bytes && best == p->rank_bytes) best = match_len(d, i, i - bdist, cmplen);
        if (p->far4_dist && best == 4 && bdist > p->far4_dist) best = 0;
        if (best >= MM) { len16[i] = (uint16_t)best; dist16[i] = (uint16_t)bdist; }
      }
      if (p->stride2) {
        for (uint32_t i = b; i < e; i++) {
          if (!(i & 1) || i + 1 >= n || (i + 1) % R == 0) continue;
          if (i + 1 >= e) continue;
          uint32_t l = len16[i + 1], dd = dist16[i + 1];
          if (l && dd <= i && d[i] == d[i - dd]) {
            uint32_t nl = l + 1;
            if (p->cap && nl > p->cap) nl = p->cap;
            if (nl > 258) nl = 258;
            len16[i] = (uint16_t)nl;
            dist16[i] = (uint16_t)dd;
          }
        }
      }
    }
    return;
  }

  uint32_t *T = m->T, *far = m->far;
  const uint32_t xs2 = p->stride2 ? 2u : p->x_stride2; /* stride2 = the analysis knob's value 2: even positions searched */
  for (uint32_t s = s0; s < s1; s++) {
    uint32_t b = s * W;
    if (b >= n) break;
    uint32_t e = b + W < n ? b + W : n;
    /* read phase */
    for (uint32_t i = b; i < e; i++)
      for (uint32_t t = 0; t < NT; t++) {
        uint32_t need = t ? LB : MM;
        for (uint32_t k = 0; k < D; k++) far[((i - b) * NT + t) * D + k] = 0;
        if (i + need > n) continue;
        uint32_t h = t ? hash_long(d + i, p) : hash_of(load32(d + i), p);
        for (uint32_t k = 0; k < D; k++) far[((i - b) * NT + t) * D + k] = T[(t * D + k) * HS + h];
      }
    /* insert phase: level 0 by MAX, deeper levels take what the level above held */
    for (uint32_t i = b; i < e; i++)
      for (uint32_t t = 0; t < NT; t++) {
        uint32_t need = t ? LB : MM;
        if (i + need > n) continue;
        uint32_t h = t ? hash_long(d + i, p) : hash_of(load32(d + i), p);
        uint32_t v = ((s + 1) << 12) | (4095 - (i - b));
        for (uint32_t k = D - 1; k > 0; k--) T[(t * D + k) * HS + h] = far[((i - b) * NT + t) * D + k - 1];
        uint32_t* slot = &T[(t * D) * HS + h];
        if (v > *slot) *slot = v;
      }
    /* candidate evaluation */
    for (uint32_t i = b; i < e; i++) {
      uint32_t shift = 0; /* analysis knob: only one parity is searched ... */
      if (xs2 && (i & 1) != (xs2 & 1)) {
        if (!(xs2 & 4) || i + 1 >= e) continue;
        shift = 1; /* ... the other tries its successor's candidates, moved back by one */
      }
      uint32_t rend = (i / R + 1) * R;
      uint32_t maxlen = n - i < 258 ? n - i : 258;
      if (rend - i < maxlen) maxlen = rend - i;
      uint32_t cmplen = (p->cap && p->cap < maxlen) ? p->cap : maxlen;
      uint32_t best = 0, bdist = 0;
      if (p->use_prev && i >= 1) { /* distance 1: ranked like the others; it wins every tie */
        uint32_t l = match_len(d, i, i - 1, cmplen);
        if (p->rank_bytes && l > p->rank_bytes) l = p->rank_bytes;
        if (l) { best = l; bdist = 1; }
      }
      for (uint32_t t = 0; t < NT; t++) {
        uint32_t need = t ? LB : MM;
        if (i + shift + need > n) continue;
        uint32_t h = t ? hash_long(d + i + shift, p) : hash_of(load32(d + i + shift), p);
        uint32_t cand[8], nc = 0;
        if (t ? p->x_long_near : p->use_near) {
          uint32_t c = ent_pos(T[(t * D) * HS + h], W);
          if (c < i + shift && c >= shift) cand[nc++] = c - shift;
        }
        for (uint32_t k = 0; k < (t && p->x_long_levels ? p->x_long_levels : D); k++) {
          uint32_t v = far[((i + shift - b) * NT + t) * D + k];
          if (v && ent_pos(v, W) >= shift) cand[nc++] = ent_pos(v, W) - shift;
        }
        for (uint32_t k = 0; k < nc; k++) {
          uint32_t dist = i - cand[k];
          if (dist > (p->x_window ? p->x_window : SFO_WINDOW)) continue;
          uint32_t l = match_len(d, i, cand[k], cmplen);
          if (p->rank_bytes && l > p->rank_bytes) l = p->rank_bytes;
          if (l > best || (l == best && l && dist < bdist)) { best = l; bdist = dist; }
        }
      }
      if (p->rank_bytes && best == p->rank_bytes) best = match_len(d, i, i - bdist, cmplen);
      /* a 4-byte match far away costs more bits than four literals (length code + 5-bit
       * distance code + up to 13 extra bits): drop it */
      if (p->far4_dist && best == 4 && bdist > p->far4_dist) best = 0;
      if (best >= MM) { len16[i] = (uint16_t)best; dist16[i] = (uint16_t)bdist; }
    }
    if (xs2 & 8) {
      /* analysis: the other parity inherits its PREDECESSOR's match, one byte shorter */
      for (uint32_t i = b; i < e; i++) {
        if ((i & 1) == (xs2 & 1) || i == b || i % R == 0) continue;
        uint32_t l = len16[i - 1], dd = dist16[i - 1];
        if (l >= MM + 1) {
          uint32_t nl = l - 1;
          if (p->far4_dist && nl == 4 && dd > p->far4_dist) continue;
          len16[i] = (uint16_t)nl;
          dist16[i] = (uint16_t)dd;
        }
      }
    } else if (xs2 && !(xs2 & 4)) {
      /* the other parity inherits its successor's match, one byte longer, when the byte before it matches too */
      for (uint32_t i = b; i < e; i++) {
        if ((i & 1) == (xs2 & 1) || i + 1 >= n || (i + 1) % R == 0) continue;
        if (i + 1 >= e) continue; /* successor belongs to the next step: not known yet */
        uint32_t l = len16[i + 1], dd = dist16[i + 1];
        if (l && dd <= i && d[i] == d[i - dd]) {
          uint32_t nl = l + 1;
          if (p->cap && nl > p->cap) nl = p->cap;
          if (nl > 258) nl = 258; /* (without a cap: the format's longest match) */
          len16[i] = (uint16_t)nl;
          dist16[i] = (uint16_t)dd;
        }
      }
    }
  }
}

/* Stage parse for the regions [r0, r1) of a strip: greedy with lazy deferral, independent per region.
 * literal_only: emit every position as a literal (stored fast path).  Returns the tokens written. */
static uint32_t parse_regions(const uint8_t* data, uint32_t n, const sfo_params* p, const uint16_t* len16,
                              const uint16_t* dist16, uint32_t r0, uint32_t r1, int literal_only,
                              uint32_t* tokens, uint32_t* ntok) {
  const uint32_t R = p->region_bytes, MM = p->min_match;
  uint32_t total = 0;
  for (uint32_t r = r0; r < r1; r++) {
    uint32_t pos = r * R, end = pos + R < n ? pos + R : n, k = 0;
    uint32_t* out = tokens + (size_t)r * R;
    if (literal_only) {
      while (pos < end) out[k++] = data[pos++];
      ntok[r] = k;
      total += k;
      continue;
    }
    while (pos < end) {
      uint32_t l = len16[pos];
      int take = l >= MM;
      /* lazy deferral, up to p->lazy positions ahead: a match is not taken when a position k
       * ahead (inside the region) offers one longer than l + (k-1) */
      for (uint32_t k2 = 1; take && k2 <= p->lazy; k2++)
        if (pos + k2 < end && len16[pos + k2] > l + (k2 - 1)) take = 0;
      if (take) {
        uint32_t dist = dist16[pos];
        if (p->cap && l >= p->cap) { /* capped at match time: extend at the chain position */
          uint32_t maxlen = end - pos < 258 ? end - pos : 258;
          uint32_t c = pos - dist;
          while (l < maxlen && data[pos + l] == data[c + l]) l++;
          /* a long match whose bytes all equal the byte before it (a run of one byte value) is coded at
           * distance 1 -- an overlapping copy, /root/reference/src/decompress.cpp:388-398 -- with the run's
           * length.  Only matches of at least SFO_RUN_MIN bytes with room to be longer are looked at (the
           * ones a whole wave extends on the GPU) */
          if (p->run_dist1 && pos >= 1 && maxlen > SFO_RUN_MIN && l >= SFO_RUN_MIN) {
            uint32_t r = 0;
            while (r < maxlen && data[pos + r] == data[pos - 1]) r++;
            if (r >= l) { l = r; dist = 1; }
          }
        }
        out[k++] = SFO_TOK_MATCH | ((l - 3) << 16) | (uint32_t)(dist - 1);
        pos += l;
      } else {
        out[k++] = data[pos];
        pos += 1;
      }
    }
    ntok[r] = k;
    total += k;
  }
  return total;
}

static uint32_t est_log2(uint32_t x);

/* Stored BY THE PROBE (round 6).  A full block that takes the stored fast path (below) and whose first SFO_SKIP_SPAN BYTES are
 * as good as uniform -- their entropy in the plan's fixed point (est_log2) within SFO_STORE_MARGIN bytes of SFO_SKIP_SPAN --
 * is stored outright when the block type is the encoder's to choose (strategy 0): it has NO tokens at all, and
 * sfo_plan_chunk stores a block without tokens.  What the rest of the block holds is then never looked at (on the GPU:
 * never fetched by the match kernel): a block whose probe span is noise and whose rest is not loses what a Huffman code
 * over its literals would have saved -- matches it had given up already. */
static int probe_span_is_noise(const uint8_t* d) {
  uint32_t f[256] = {0}, ent = 0;
  for (uint32_t i = 0; i < SFO_SKIP_SPAN; i++) f[d[i]]++;
  for (uint32_t s = 0; s < 256; s++)
    if (f[s]) ent += f[s] * (est_log2(SFO_SKIP_SPAN) - est_log2(f[s]));
  return ((ent >> 8) + 7) / 8 + SFO_STORE_MARGIN >= SFO_SKIP_SPAN;
}

/*
 * Stages n1 + parse for one strip, DEFLATE block by block (chunk_bytes each): the hash tables and the
 * window carry over from block to block.
 * Stored fast path: when the first SFO_SKIP_SPAN positions of a block parse to (almost) nothing but
 * literals -- at least SFO_SKIP_SPAN - SFO_SKIP_SLACK tokens -- the rest of the block is neither
 * searched nor inserted into the tables, and EVERY position of the block is emitted as a literal.  (High-entropy
 * data then costs a quarter of the match work and ends up in a stored block.)  Round 5: the block BEHIND such a block
 * in its strip is probed on SFO_SKIP_PROBE positions only (a sixteenth of the match work while the data stays
 * high-entropy).
 * tokens: region r's tokens at tokens[r * region_bytes + k], k < ntok[r] (regions counted from the
 * strip's start); a block's regions are those it covers.
 */
int sfo_strip_tokens(const uint8_t* src, uint32_t n, const sfo_params* p, uint32_t* tokens, uint32_t* ntok) {
  const uint32_t cb = p->chunk_bytes, R = p->region_bytes, W = p->step;
  matcher m;
  if (matcher_init(&m, src, n, p)) { matcher_free(&m); return -3; }
  int prev_skipped = 0; /* the strip's previous block took the stored fast path */
  for (uint32_t c0 = 0; c0 < n; c0 += cb) {
    const uint32_t cn = n - c0 < cb ? n - c0 : cb;
    const uint32_t r0 = c0 / R, r1 = (c0 + cn + R - 1) / R;
    const uint32_t s0 = c0 / W, s1 = (c0 + cn + W - 1) / W;
    const int can_skip = p->fast_skip && cn > SFO_SKIP_SPAN && SFO_SKIP_SPAN % R == 0 && SFO_SKIP_SPAN % W == 0;
    if (!can_skip) {
      match_steps(&m, s0, s1);
      parse_regions(m.d, n, p, m.len16, m.dist16, r0, r1, 0, tokens, ntok);
      prev_skipped = 0;
      continue;
    }
    /* Behind a block that took the fast path only the first SFO_SKIP_PROBE positions of the probe span are searched and
     * inserted (the rest of the span are literals whatever they hold): high-entropy data usually goes on.  The rule that
     * decides stays the same -- tokens of the whole span -- so such a block needs nearly all of the searched positions
     * to be literals as well. */
    const uint32_t probe = (prev_skipped && SFO_SKIP_PROBE % W == 0) ? SFO_SKIP_PROBE : SFO_SKIP_SPAN;
    const uint32_t sh = s0 + SFO_SKIP_SPAN / W, rh = r0 + SFO_SKIP_SPAN / R;
    match_steps(&m, s0, s0 + probe / W);
    const uint32_t head = parse_regions(m.d, n, p, m.len16, m.dist16, r0, rh, 0, tokens, ntok);
    const int skip = head >= SFO_SKIP_SPAN - SFO_SKIP_SLACK;
    if (skip && p->strategy == 0 && cn == cb && probe_span_is_noise(src + c0)) {  /* stored by the probe: no tokens */
      for (uint32_t r = r0; r < r1; r++) ntok[r] = 0;
      prev_skipped = 1;
      continue;
    }
    /* (round 5) the WHOLE block is literals then, the probe span included: its few matches are dropped */
    if (skip) parse_regions(m.d, n, p, m.len16, m.dist16, r0, rh, 1, tokens, ntok);
    if (!skip) match_steps(&m, sh, s1);
    parse_regions(m.d, n, p, m.len16, m.dist16, rh, r1, skip, tokens, ntok);
    prev_skipped = skip;
  }
  matcher_free(&m);
  return 0;
}

/* one independent chunk (a strip of its own): stage n1 alone, then the parse alone */
void sfo_match_chunk(const uint8_t* src, uint32_t n, const sfo_params* p, uint16_t* len16,
                     uint16_t* dist16) {
  matcher m;
  memset(len16, 0, (size_t)n * 2);
  memset(dist16, 0, (size_t)n * 2);
  if (matcher_init(&m, src, n, p) == 0) {
    match_steps(&m, 0, (n + p->step - 1) / p->step);
    memcpy(len16, m.len16, (size_t)n * 2);
    memcpy(dist16, m.dist16, (size_t)n * 2);
  }
  matcher_free(&m);
}

void sfo_parse_chunk(const uint8_t* data, uint32_t n, const sfo_params* p,
                     const uint16_t* len16, const uint16_t* dist16, uint32_t* tokens,
                     uint32_t* ntok) {
  const uint32_t R = p->region_bytes;
  const uint32_t nreg = (n + R - 1) / R;
  const int can_skip = p->fast_skip && n > SFO_SKIP_SPAN && SFO_SKIP_SPAN % R == 0;
  if (!can_skip) {
    parse_regions(data, n, p, len16, dist16, 0, nreg, 0, tokens, ntok);
    return;
  }
  const uint32_t rh = SFO_SKIP_SPAN / R;
  const uint32_t head = parse_regions(data, n, p, len16, dist16, 0, rh, 0, tokens, ntok);
  const int skip = head >= SFO_SKIP_SPAN - SFO_SKIP_SLACK;
  if (skip && p->strategy == 0 && n == p->chunk_bytes && probe_span_is_noise(data)) {
    for (uint32_t r = 0; r < nreg; r++) ntok[r] = 0;
    return;
  }
  if (skip) parse_regions(data, n, p, len16, dist16, 0, rh, 1, tokens, ntok);
  parse_regions(data, n, p, len16, dist16, rh, nreg, skip, tokens, ntok);
}

static inline uint32_t len_symbol(uint32_t len) { /* 3..258 -> 257..285 */
  uint32_t s = 28;
  if (len == 258) return 285;
  while (len_base[s] > len) s--;
  return 257 + s;
}
static inline uint32_t dist_symbol(uint32_t dist) { /* 1..32768 -> 0..29 */
  uint32_t s = 29;
  while (dist_base[s] > dist) s--;
  return s;
}

void sfo_histogram(const uint32_t* tokens, const uint32_t* ntok, uint32_t nregions,
                   uint32_t region_bytes, uint32_t* ll, uint32_t* d) {
  memset(ll, 0, 286 * 4);
  memset(d, 0, 30 * 4);
  for (uint32_t r = 0; r < nregions; r++)
    for (uint32_t k = 0; k < ntok[r]; k++) {
      uint32_t t = tokens[(size_t)r * region_bytes + k];
      if (t & SFO_TOK_MATCH) {
        ll[len_symbol(((t >> 16) & 0xFF) + 3)]++;
        d[dist_symbol((t & 0x7FFF) + 1)]++;
      } else {
        ll[t & 0xFF]++;
      }
    }
  ll[256] = 1;
}

/*
 * Stage n3.  Huffman tree by the two-queue method over symbols sorted ascending
 * by (freq, symbol) (ties: a leaf is taken before an internal node of equal
 * weight), leaf depths counted per length with depths beyond maxbits clamped;
 * the Kraft excess of the clamp is removed by lengthening the deepest leaves
 * still shorter than maxbits, an overshoot is returned by shortening maxbits
 * leaves; lengths are then dealt out longest-first to the rarest symbols.
 * One used symbol gets a 1-bit code (reference: table.hpp:116-120; decoder
 * accepts it, SURVEY.md 8(c) probe B); zero used symbols -> all lengths 0.
 */
void sfo_build_lengths(const uint32_t* freq, uint32_t n, uint32_t maxbits, uint8_t* lens) {
  uint32_t key[288], m = 0;
  memset(lens, 0, n);
  for (uint32_t s = 0; s < n; s++)
    if (freq[s]) key[m++] = (freq[s] << 9) | s; /* freq <= 2^16+, sym < 512 */
  if (m == 0) return;
  if (m == 1) { lens[key[0] & 511] = 1; return; }
  /* sort ascending (insertion; m <= 286) */
  for (uint32_t i = 1; i < m; i++) {
    uint32_t k = key[i], j = i;
    while (j > 0 && key[j - 1] > k) { key[j] = key[j - 1]; j--; }
    key[j] = k;
  }
  uint32_t w[576], parent[576], depth[576];
  for (uint32_t i = 0; i < m; i++) w[i] = key[i] >> 9;
  uint32_t i = 0, j = m, k = m;
  while (k < 2 * m - 1) {
    uint32_t a, b;
    if (i < m && (j >= k || w[i] <= w[j])) a = i++; else a = j++;
    if (i < m && (j >= k || w[i] <= w[j])) b = i++; else b = j++;
    w[k] = w[a] + w[b];
    parent[a] = parent[b] = k;
    k++;
  }
  uint32_t cnt[17] = {0};
  depth[2 * m - 2] = 0;
  for (int32_t v = (int32_t)(2 * m - 3); v >= 0; v--) {
    depth[v] = depth[parent[v]] + 1;
    if ((uint32_t)v < m) cnt[depth[v] < maxbits ? depth[v] : maxbits]++;
  }
  int64_t over = -((int64_t)1 << maxbits);
  for (uint32_t l = 1; l <= maxbits; l++) over += (int64_t)cnt[l] << (maxbits - l);
  while (over > 0) {
    uint32_t l = maxbits - 1;
    while (cnt[l] == 0) l--;
    cnt[l]--;
    cnt[l + 1]++;
    over -= (int64_t)1 << (maxbits - l - 1);
  }
  while (over < 0) {
    cnt[maxbits]--;
    cnt[maxbits - 1]++;
    over++;
  }
  uint32_t idx = 0;
  for (uint32_t l = maxbits; l >= 1; l--)
    for (uint32_t c = 0; c < cnt[l]; c++) lens[key[idx++] & 511] = (uint8_t)l;
}

/* LSB-first bit writer (inverse of bit_span / pop_bits) */
typedef struct {
  uint8_t* p;
  size_t bitpos;
} bitw;
static void put_bits(bitw* w, uint32_t v, uint32_t n) {
  for (uint32_t i = 0; i < n; i++, w->bitpos++)
    if ((v >> i) & 1) w->p[w->bitpos >> 3] |= (uint8_t)(1u << (w->bitpos & 7));
}
static uint32_t rev_bits(uint32_t v, uint32_t n) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1) << (n - 1 - i);
  return r;
}

static uint32_t fixed_ll_len(uint32_t s) { return s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8; }

/* RLE of one code-length sequence (never crosses into the other sequence and
 * never starts with 16: src/decompress.cpp:253-312 keeps per-sequence state). */
static uint32_t rle_lengths(const uint8_t* lens, uint32_t n, uint8_t* sym, uint8_t* ext) {
  uint32_t k = 0, i = 0;
  while (i < n) {
    uint8_t v = lens[i];
    uint32_t r = 1;
    while (i + r < n && lens[i + r] == v) r++;
    i += r;
    if (v == 0) {
      while (r >= 11) { uint32_t c = r < 138 ? r : 138; sym[k] = 18; ext[k++] = (uint8_t)(c - 11); r -= c; }
      if (r >= 3) { sym[k] = 17; ext[k++] = (uint8_t)(r - 3); r = 0; }
      while (r--) { sym[k] = 0; ext[k++] = 0; }
    } else {
      sym[k] = v; ext[k++] = 0; r--;
      while (r >= 3) { uint32_t c = r < 6 ? r : 6; sym[k] = 16; ext[k++] = (uint8_t)(c - 3); r -= c; }
      while (r--) { sym[k] = v; ext[k++] = 0; }
    }
  }
  return k;
}

/* ~ 256 * log2(x), x >= 1: the exponent and the top six mantissa bits through a table (integers only: the GPU's k_plan
 * computes the same) */
static const uint8_t est_lg64[64] = {0, 6, 11, 17, 22, 28, 33, 38, 44, 49, 54, 59, 63, 68, 73, 78, 82, 87, 92, 96, 100, 105, 109, 113, 118, 122,
                                     126, 130, 134, 138, 142, 146, 150, 154, 157, 161, 165, 169, 172, 176, 179, 183, 186, 190, 193, 197,
                                     200, 203, 207, 210, 213, 216, 220, 223, 226, 229, 232, 235, 238, 241, 244, 247, 250, 253};
static uint32_t est_log2(uint32_t x) {
  uint32_t e = 0;
  for (uint32_t v = x | 1u; v > 1; v >>= 1) e++;
  uint32_t m = (e >= 6 ? x >> (e - 6) : x << (6 - e)) & 63u;
  return (e << 8) + est_lg64[m];
}

void sfo_plan_chunk(const uint32_t* ll, const uint32_t* d, uint32_t n_raw, int is_last,
                    const sfo_params* p, sfo_plan* plan) {
  memset(plan, 0, sizeof *plan);
  if (p->strategy == 0) {
    /* Stored without a code (round 5).  A chunk that is (all but) incompressible -- the fixed block no shorter than the
     * stored one, and the ESTIMATE of the dynamic block (entropy of the symbols in fixed point + extra bits + the
     * shortest header) within SFO_STORE_MARGIN bytes of it -- is stored, and no Huffman code is built for it: such a
     * chunk gives up SFO_STORE_MARGIN bytes at most, and the high-entropy chunks of the stored fast path cost the
     * planner a histogram pass. */
    uint32_t tot = 0, nmat = 0, ent = 0, extra = 0, fixb = 0;
    for (uint32_t s = 0; s < 286; s++) tot += ll[s];
    if (n_raw && tot == 1) { /* bytes but no tokens: stored by the probe (sfo_strip_tokens) */
      plan->btype = 0;
      plan->out_bytes = n_raw + 5;
      return;
    }
    for (uint32_t s = 0; s < 30; s++) nmat += d[s];
    for (uint32_t s = 0; s < 286; s++) {
      if (ll[s]) ent += ll[s] * (est_log2(tot) - est_log2(ll[s]));
      fixb += ll[s] * fixed_ll_len(s);
      if (s >= 257) extra += ll[s] * len_extra[s - 257];
    }
    for (uint32_t s = 0; s < 30; s++) {
      if (d[s]) ent += d[s] * (est_log2(nmat) - est_log2(d[s]));
      extra += d[s] * dist_extra[s];
    }
    const int fin0 = is_last && p->final_stream;
    const uint32_t est_bits = (ent >> 8) + extra + SFO_EST_HEADER_BITS, fixbits = 3 + fixb + extra + 5 * nmat;
    const uint32_t est_b = fin0 ? (est_bits + 7) / 8 : (est_bits + 3 + 7) / 8 + 4;
    const uint32_t fix_b = fin0 ? (fixbits + 7) / 8 : (fixbits + 3 + 7) / 8 + 4;
    if (fix_b >= n_raw + 5 && est_b + SFO_STORE_MARGIN >= n_raw + 5) {
      plan->btype = 0;
      plan->out_bytes = n_raw + 5;
      return;
    }
  }
  sfo_build_lengths(ll, 286, 15, plan->ll_lens);
  sfo_build_lengths(d, 30, 15, plan->d_lens);

  uint32_t extra = 0, nmatch = 0, dyn_body = 0, fix_body = 0;
  for (uint32_t s = 0; s < 286; s++) {
    dyn_body += ll[s] * plan->ll_lens[s];
    fix_body += ll[s] * fixed_ll_len(s);
    if (s >= 257) extra += ll[s] * len_extra[s - 257];
  }
  for (uint32_t s = 0; s < 30; s++) {
    dyn_body += d[s] * plan->d_lens[s];
    extra += d[s] * dist_extra[s];
    nmatch += d[s];
  }
  dyn_body += extra;
  fix_body += extra + 5 * nmatch;

  /* dynamic header */
  uint32_t hlit = 286, hdist = 30;
  while (hlit > 257 && plan->ll_lens[hlit - 1] == 0) hlit--;
  while (hdist > 1 && plan->d_lens[hdist - 1] == 0) hdist--;
  uint8_t sym[320], ext[320];
  uint32_t nl = rle_lengths(plan->ll_lens, hlit, sym, ext);
  uint32_t nd = rle_lengths(plan->d_lens, hdist, sym + nl, ext + nl);
  uint32_t clf[19] = {0};
  for (uint32_t k = 0; k < nl + nd; k++) clf[sym[k]]++;
  uint8_t cll[19];
  sfo_build_lengths(clf, 19, 7, cll);
  uint32_t hclen = 19;
  while (hclen > 4 && cll[cl_order[hclen - 1]] == 0) hclen--;
  uint32_t clc[19];
  sfo_canonical_codes(cll, 19, clc);
  bitw w = {plan->header, 0};
  put_bits(&w, hlit - 257, 5);
  put_bits(&w, hdist - 1, 5);
  put_bits(&w, hclen - 4, 4);
  for (uint32_t k = 0; k < hclen; k++) put_bits(&w, cll[cl_order[k]], 3);
  for (uint32_t k = 0; k < nl + nd; k++) {
    put_bits(&w, rev_bits(clc[sym[k]], cll[sym[k]]), cll[sym[k]]);
    if (sym[k] == 16) put_bits(&w, ext[k], 2);
    else if (sym[k] == 17) put_bits(&w, ext[k], 3);
    else if (sym[k] == 18) put_bits(&w, ext[k], 7);
  }
  plan->header_bits = (uint32_t)w.bitpos;

  uint32_t dyn_bits = 3 + plan->header_bits + dyn_body;
  uint32_t fix_bits = 3 + fix_body;
  /* only the final block of a final stream may end unaligned; every other Huffman
   * block is followed by an empty stored block so the next chunk / shard starts on a byte */
  int fin = is_last && p->final_stream;
  uint32_t dyn_bytes = fin ? (dyn_bits + 7) / 8 : (dyn_bits + 3 + 7) / 8 + 4;
  uint32_t fix_bytes = fin ? (fix_bits + 7) / 8 : (fix_bits + 3 + 7) / 8 + 4;
  uint32_t sto_bytes = n_raw + 5;
  uint32_t bt;
  if (p->strategy == 1) bt = 0;
  else if (p->strategy == 2) bt = 1;
  else if (p->strategy == 3) bt = 2;
  else {
    bt = 0;
    uint32_t best = sto_bytes;
    if (fix_bytes < best) { bt = 1; best = fix_bytes; }
    if (dyn_bytes < best) { bt = 2; best = dyn_bytes; }
  }
  plan->btype = bt;
  plan->out_bytes = bt == 0 ? sto_bytes : bt == 1 ? fix_bytes : dyn_bytes;
  plan->body_bits = bt == 1 ? fix_body : dyn_body;
}

/* Stage n4 for one chunk; dst zero-initialised by the caller. */
/* sub (NULL or SFO_SUB_REGIONS pairs): per parse region the bit offset, from the chunk's first byte, of the
 * region's first token code, and the number of tokens before it; regions past the data name the
 * end-of-block code and the token total; a stored chunk has all zeros. */
static void emit_chunk(const uint8_t* data, uint32_t n, const uint32_t* tokens,
                       const uint32_t* ntok, const sfo_params* p, const sfo_plan* plan,
                       int is_last, uint8_t* dst, uint32_t* sub) {
  int bfinal = is_last && p->final_stream;
  if (sub) memset(sub, 0, 2 * SFO_SUB_REGIONS * sizeof(uint32_t));
  if (plan->btype == 0) {
    dst[0] = (uint8_t)bfinal; /* BFINAL, BTYPE=00, 5 pad bits */
    dst[1] = (uint8_t)(n & 0xFF);
    dst[2] = (uint8_t)(n >> 8);
    dst[3] = (uint8_t)(~n & 0xFF);
    dst[4] = (uint8_t)((~n >> 8) & 0xFF);
    memcpy(dst + 5, data, n);
    return;
  }
  bitw w = {dst, 0};
  put_bits(&w, (uint32_t)bfinal, 1);
  put_bits(&w, plan->btype, 2);
  uint8_t ll_lens[288], d_lens[32];
  if (plan->btype == 2) {
    for (uint32_t k = 0; k < plan->header_bits; k++)
      put_bits(&w, (plan->header[k >> 3] >> (k & 7)) & 1, 1);
    memcpy(ll_lens, plan->ll_lens, 288);
    memcpy(d_lens, plan->d_lens, 32);
  } else {
    for (uint32_t s = 0; s < 288; s++) ll_lens[s] = (uint8_t)fixed_ll_len(s);
    for (uint32_t s = 0; s < 32; s++) d_lens[s] = 5;
  }
  uint32_t llc[288], dc[32];
  sfo_canonical_codes(ll_lens, 288, llc);
  sfo_canonical_codes(d_lens, 32, dc);
  uint32_t nreg = (n + p->region_bytes - 1) / p->region_bytes, seen = 0, nsub = 0;
  for (uint32_t r = 0; r < nreg; r++) {
    /* sub-index entry per SFO_SUB_BYTES of input (a region boundary, hence a token start) */
    if (sub && (r * p->region_bytes) % SFO_SUB_BYTES == 0 && nsub < SFO_SUB_REGIONS) {
      sub[2 * nsub] = (uint32_t)w.bitpos;
      sub[2 * nsub + 1] = seen;
      nsub++;
    }
    seen += ntok[r];
    for (uint32_t k = 0; k < ntok[r]; k++) {
      uint32_t t = tokens[(size_t)r * p->region_bytes + k];
      if (t & SFO_TOK_MATCH) {
        uint32_t len = ((t >> 16) & 0xFF) + 3, dist = (t & 0x7FFF) + 1;
        uint32_t ls = len_symbol(len), ds = dist_symbol(dist);
        put_bits(&w, rev_bits(llc[ls], ll_lens[ls]), ll_lens[ls]);
        put_bits(&w, len - len_base[ls - 257], len_extra[ls - 257]);
        put_bits(&w, rev_bits(dc[ds], d_lens[ds]), d_lens[ds]);
        put_bits(&w, dist - dist_base[ds], dist_extra[ds]);
      } else {
        put_bits(&w, rev_bits(llc[t], ll_lens[t]), ll_lens[t]);
      }
    }
  }
  for (uint32_t r = nsub; sub && r < SFO_SUB_REGIONS; r++) { sub[2 * r] = (uint32_t)w.bitpos; sub[2 * r + 1] = seen; }
  put_bits(&w, rev_bits(llc[256], ll_lens[256]), ll_lens[256]);
  if (!bfinal) {
    /* byte-align with an empty non-final stored block: 000, pad, 00 00 FF FF */
    put_bits(&w, 0, 3);
    w.bitpos = (w.bitpos + 7) & ~(size_t)7;
    size_t o = w.bitpos >> 3;
    dst[o] = 0; dst[o + 1] = 0; dst[o + 2] = 0xFF; dst[o + 3] = 0xFF;
  }
}

/* strip_bytes = 0: the largest power-of-two multiple of chunk_bytes up to SFO_DEFAULT_STRIP chunks' worth (SFO_CHAIN_STRIP
 * with hash chains) that still gives SFO_MIN_STRIPS strips (a function of n and of the matcher alone; the GPU library uses
 * the same rule for block_bytes = 0) */
size_t sfo_resolve_strip_bytes(const sfo_params* p, size_t n) {
  if (p->strip_bytes) return p->strip_bytes;
  const size_t base = (size_t)p->chunk_bytes * SFO_DEFAULT_STRIP_CHUNKS;
  size_t b = (size_t)p->chunk_bytes * (p->chain_depth ? SFO_CHAIN_STRIP_CHUNKS : SFO_LARGE_STRIP_CHUNKS);
  while (b > base && n / b < (p->chain_depth ? SFO_CHAIN_MIN_STRIPS : SFO_LARGE_MIN_STRIPS)) b >>= 1;
  while (b > p->chunk_bytes && n / b < SFO_MIN_STRIPS) b >>= 1;
  return b;
}

size_t sfo_compress_bound(size_t n, const sfo_params* p) {
  size_t cb = p->chunk_bytes;
  size_t nchunks = n ? (n + cb - 1) / cb : 1;
  return nchunks * (cb + cb / 8 + 640); /* the per-chunk slack also covers the <= 18 wrapper bytes */
}

/* RFC 1952 section 8: CRC-32, reflected polynomial 0xEDB88320, one bit at a time */
uint32_t sfo_crc32(const uint8_t* data, size_t n) {
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) {
    c ^= data[i];
    for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
  }
  return ~c;
}

/* RFC 1950 section 8.2: s1 = 1 + sum of bytes, s2 = sum of the s1 values, both mod 65521 */
uint32_t sfo_adler32(const uint8_t* data, size_t n) {
  uint32_t s1 = 1, s2 = 0;
  for (size_t i = 0; i < n; i++) {
    s1 = (s1 + data[i]) % 65521u;
    s2 = (s2 + s1) % 65521u;
  }
  return (s2 << 16) | s1;
}

/* crc(A||B): append len_b zero bytes to A's register (bit-serially squared operator, here simply
 * by feeding zero bits through the shift register in log steps) and xor crc(B).
 * Uses: crc(A||B) = crc(A||0^len) ^ crc(0^len) ^ crc(B)  (linearity over GF(2)). */
static uint32_t gf2_mul(uint32_t a, uint32_t b) { /* a(x)*b(x) mod P, reflected bit order, x^0 = bit 31 */
  uint32_t prod = 0;
  for (int k = 31; k >= 0; k--) {
    if ((a >> k) & 1) prod ^= b;
    b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
  }
  return prod;
}

uint32_t sfo_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  /* x^(8*len_b) by square and multiply; x^8 is bit 23 in this representation */
  uint32_t op = 0x80000000u, base = 0x00800000u;
  for (uint64_t e = len_b; e; e >>= 1) {
    if (e & 1) op = gf2_mul(op, base);
    base = gf2_mul(base, base);
  }
  return gf2_mul(op, crc_a) ^ crc_b;
}

uint32_t sfo_adler32_combine(uint32_t adler_a, uint32_t adler_b, uint64_t len_b) {
  const uint64_t M = 65521u;
  uint64_t a1 = adler_a & 0xFFFF, b1 = adler_a >> 16, a2 = adler_b & 0xFFFF, b2 = adler_b >> 16;
  uint64_t a = (a1 + a2 + M - 1) % M;
  uint64_t b = (b1 + b2 + (len_b % M) * ((a1 + M - 1) % M)) % M;
  return (uint32_t)((b << 16) | a);
}

int sfo_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len,
                 const sfo_params* p) {
  return sfo_compress_indexed(src, n, dst, cap, out_len, p, NULL, NULL);
}

int sfo_compress_indexed(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len,
                         const sfo_params* p, uint64_t* index, uint32_t* subindex) {
  if (p->container) {
    if (p->container > 2 || !p->final_stream) return -1;
    static const uint8_t zhdr[2] = {0x78, 0x9C};
    static const uint8_t ghdr[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 0xFF};
    const size_t h = p->container == 1 ? 2 : 10, t = p->container == 1 ? 4 : 8;
    if (cap < h + t) return -2;
    sfo_params raw = *p;
    raw.container = 0;
    size_t body = 0;
    int rc = sfo_compress_indexed(src, n, dst + h, cap - h - t, &body, &raw, index, subindex);
    if (rc) return rc;
    if (index) {
      size_t nch = n ? (n + p->chunk_bytes - 1) / p->chunk_bytes : 1;
      for (size_t c = 0; c <= nch; c++) index[c] += h; /* offsets into the wrapped stream */
    }
    memcpy(dst, p->container == 1 ? zhdr : ghdr, h);
    uint8_t* tr = dst + h + body;
    if (p->container == 1) {
      uint32_t a = sfo_adler32(src, n);
      tr[0] = (uint8_t)(a >> 24); tr[1] = (uint8_t)(a >> 16); tr[2] = (uint8_t)(a >> 8); tr[3] = (uint8_t)a;
    } else {
      uint32_t c = sfo_crc32(src, n), isz = (uint32_t)n;
      for (int k = 0; k < 4; k++) { tr[k] = (uint8_t)(c >> (8 * k)); tr[4 + k] = (uint8_t)(isz >> (8 * k)); }
    }
    *out_len = h + body + t;
    return 0;
  }
  if (p->chunk_bytes == 0 || p->chunk_bytes > 32768 || p->region_bytes == 0 ||
      p->step == 0 || p->step > 4096 || p->hash_bits < 8 || p->hash_bits > 16 ||
      (p->min_match != 3 && p->min_match != 4))
    return -1;
  const size_t cb = p->chunk_bytes;
  const size_t sb = sfo_resolve_strip_bytes(p, n);
  /* a strip is a whole number of DEFLATE blocks; blocks, regions and steps nest */
  if (sb % cb || sb > (1u << 28) || (sb > cb && (cb % p->region_bytes || cb % p->step))) return -1;
  const size_t nchunks = n ? (n + cb - 1) / cb : 1;
  uint32_t* tokens = (uint32_t*)malloc((sb + p->region_bytes) * 4);
  uint32_t* ntok = (uint32_t*)calloc(sb / p->region_bytes + 2, 4);
  uint8_t* tmp = (uint8_t*)malloc(cb + cb / 8 + 1024);
  size_t off = 0;
  int rc = 0;
  if (!tokens || !ntok || !tmp) { rc = -3; goto out; }
  if ((cb + p->region_bytes - 1) / p->region_bytes > 4096) { rc = -1; goto out; }
  for (size_t s0 = 0, c = 0; c < nchunks; s0 += sb) {
    const uint32_t sn = (uint32_t)(n - s0 < sb ? n - s0 : sb);
    if ((rc = sfo_strip_tokens(src + s0, sn, p, tokens, ntok)) != 0) goto out;
    for (size_t c0 = 0; c0 < (sn ? sn : 1); c0 += cb, c++) {
      const uint8_t* data = src + s0 + c0;
      uint32_t cn = (uint32_t)(sn - c0 < cb ? sn - c0 : cb);
      int is_last = c + 1 == nchunks;
      uint32_t ll[286], d[30];
      sfo_plan plan;
      uint32_t r0 = (uint32_t)(c0 / p->region_bytes);
      uint32_t nreg = (cn + p->region_bytes - 1) / p->region_bytes;
      const uint32_t* ctok = tokens + (size_t)r0 * p->region_bytes;
      sfo_histogram(ctok, ntok + r0, nreg, p->region_bytes, ll, d);
      sfo_plan_chunk(ll, d, cn, is_last, p, &plan);
      if (off + plan.out_bytes > cap) { rc = -2; goto out; }
      memset(tmp, 0, plan.out_bytes + 8);
      if (index) index[c] = off;
      emit_chunk(data, cn, ctok, ntok + r0, p, &plan, is_last, tmp,
                 subindex ? subindex + c * 2 * SFO_SUB_REGIONS : NULL);
      memcpy(dst + off, tmp, plan.out_bytes);
      off += plan.out_bytes;
    }
  }
  *out_len = off;
  if (index) index[nchunks] = off;
out:
  free(tmp);
  free(ntok);
  free(tokens);
  return rc;
}
