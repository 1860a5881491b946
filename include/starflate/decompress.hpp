// starflate::decompress -- raw RFC 1951 inflate with the reference's signature and statuses
// (/root/reference/src/decompress.hpp:13-71).  Header-only host code over huffman::table /
// bit_span; the GPU path of this repository is compress() (compress.hpp), whose streams this
// function (and the reference's) inverts.
#pragma once
#include "starflate/compat/expected.hpp"
#include "starflate/huffman/huffman.hpp"
#include "starflate/huffman/lookup_decoder.hpp"

#include <algorithm>
#include <array>
#include <climits>
#include <cstddef>
#include <cstdint>
#include <ranges>
#include <span>
#include <utility>
#include <vector>

namespace starflate {

enum class DecompressStatus : std::uint8_t {
  Success,
  Error,  // kept for value compatibility with the reference ("TODO: remove" there); here:
          // truncated bit fields and code-length runs on which the reference asserts
  InvalidBlockHeader,
  NoCompressionLenMismatch,
  DstTooSmall,
  SrcTooSmall,
  InvalidLitOrLen,
  InvalidDistance,
};

namespace detail {

enum class BlockType : std::uint8_t { NoCompression, FixedHuffman, DynamicHuffman };

struct BlockHeader {
  bool final;
  BlockType type;
};

/// BFINAL then BTYPE, LSB first; needs 3 bits; BTYPE 3 is invalid
inline auto read_header(huffman::bit_span& bits) -> compat::expected<BlockHeader, DecompressStatus> {
  if (std::ranges::size(bits) < 3) return compat::unexpected{DecompressStatus::InvalidBlockHeader};
  const auto b = bits.begin();
  const unsigned type = unsigned{bool(b[1])} | (unsigned{bool(b[2])} << 1U);
  if (type == 3) return compat::unexpected{DecompressStatus::InvalidBlockHeader};
  const bool final = bool(b[0]);
  bits.consume(3);
  return BlockHeader{final, static_cast<BlockType>(type)};
}

/// n bytes from `distance` bytes before dst; an overlap repeats (RFC 1951 3.2.3)
inline void copy_from_before(std::uint16_t distance, std::span<std::byte>::iterator dst, std::uint16_t n) {
  auto src = dst - distance;
  for (std::uint16_t k = 0; k < n; ++k) *dst++ = *src++;
}

struct extra_base {
  std::uint8_t extra;
  std::uint16_t base;
};
// RFC 1951 3.2.5
inline constexpr std::array<extra_base, 29> kLength{{{0, 3},   {0, 4},   {0, 5},   {0, 6},   {0, 7},  {0, 8},
                                                     {0, 9},   {0, 10},  {1, 11},  {1, 13},  {1, 15}, {1, 17},
                                                     {2, 19},  {2, 23},  {2, 27},  {2, 31},  {3, 35}, {3, 43},
                                                     {3, 51},  {3, 59},  {4, 67},  {4, 83},  {4, 99}, {4, 115},
                                                     {5, 131}, {5, 163}, {5, 195}, {5, 227}, {0, 258}}};
inline constexpr std::array<extra_base, 30> kDistance{
    {{0, 1},     {0, 2},     {0, 3},      {0, 4},      {1, 5},      {1, 7},     {2, 9},     {2, 13},
     {3, 17},    {3, 25},    {4, 33},     {4, 49},     {5, 65},     {5, 97},    {6, 129},   {6, 193},
     {7, 257},   {7, 385},   {8, 513},    {8, 769},    {9, 1025},   {9, 1537},  {10, 2049}, {10, 3073},
     {11, 4097}, {11, 6145}, {12, 8193},  {12, 12289}, {13, 16385}, {13, 24577}}};
inline constexpr std::array<std::uint8_t, 19> kCodeLengthOrder{16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/// n (<= 16) bits as an LSB-first integer; nullopt when fewer remain
inline auto pop_bits(huffman::bit_span& bits, std::uint8_t n) -> std::optional<std::uint16_t> {
  if (std::ranges::size(bits) < n) return std::nullopt;
  const auto r = static_cast<std::uint16_t>(bits.peek(n));  // the reference loops over the bits (src/decompress.cpp:94-114)
  bits.consume(n);
  return r;
}

using dyn_table = huffman::table<std::uint16_t>;

inline auto fixed_tables() -> const std::pair<dyn_table, dyn_table>& {
  using span = huffman::symbol_span<std::uint16_t>;
  static const std::pair<dyn_table, dyn_table> t{
      dyn_table{huffman::symbol_bitsize,
                std::vector<std::pair<span, std::uint8_t>>{{span{0, 143}, 8}, {span{144, 255}, 9}, {span{256, 279}, 7}, {span{280, 287}, 8}}},
      dyn_table{huffman::symbol_bitsize, std::vector<std::pair<span, std::uint8_t>>{{span{0, 31}, 5}}}};
  return t;
}

/// Symbols are decoded through huffman::lookup_decoder (one array read for codes up to 9 / 7 bits, the
/// reference's per-bit walk, src/decompress.cpp:122-187 via huffman/src/decode.hpp:83-102, for the rest): same
/// results, a faster host-side checker.
template <class Table>
auto inflate_block(huffman::bit_span& bits, std::span<std::byte> dst, std::ptrdiff_t& written, const Table& lt,
                   const Table& dt) -> DecompressStatus {
  const huffman::lookup_decoder<std::uint16_t, 9> fast_ll{lt};
  const huffman::lookup_decoder<std::uint16_t, 7> fast_d{dt};
  for (;;) {
    const auto ll = fast_ll.decode_one(lt, bits);
    if (!ll.has_value()) return DecompressStatus::InvalidLitOrLen;
    bits.consume(ll.encoded_size());
    const std::uint16_t sym = ll.symbol();
    if (sym < 256) {
      if (dst.size() - static_cast<std::size_t>(written) < 1) return DecompressStatus::DstTooSmall;
      dst[static_cast<std::size_t>(written++)] = static_cast<std::byte>(sym);
      continue;
    }
    if (sym == 256) return DecompressStatus::Success;
    if (sym > 285) return DecompressStatus::InvalidLitOrLen;
    const auto& li = kLength[static_cast<std::size_t>(sym - 257)];
    const auto lx = pop_bits(bits, li.extra);
    if (!lx) return DecompressStatus::Error;
    const auto len = static_cast<std::uint16_t>(li.base + *lx);
    const auto dd = fast_d.decode_one(dt, bits);
    if (!dd.has_value()) return DecompressStatus::InvalidDistance;
    bits.consume(dd.encoded_size());
    if (dd.symbol() >= kDistance.size()) return DecompressStatus::InvalidLitOrLen;
    const auto& di = kDistance[dd.symbol()];
    const auto dx = pop_bits(bits, di.extra);
    if (!dx) return DecompressStatus::Error;
    const auto distance = static_cast<std::uint16_t>(di.base + *dx);
    if (distance > written) return DecompressStatus::InvalidDistance;
    if (dst.size() - static_cast<std::size_t>(written) < len) return DecompressStatus::DstTooSmall;
    copy_from_before(distance, dst.begin() + written, len);
    written += len;
  }
}

/// Code lengths that claim more code space than there is (Kraft sum > 1) do not describe a prefix code: the
/// reference builds a table of wrapped codes from them (an assert in debug builds, huffman/src/code.hpp), this
/// decoder reports DecompressStatus::Error -- the status it gives every input on which the reference asserts.
template <class Pairs>
auto oversubscribed(const Pairs& pairs, unsigned maxbits) -> bool {
  std::uint64_t used = 0;
  for (const auto& [sym, len] : pairs) used += std::uint64_t{1} << (maxbits - len);
  return used > (std::uint64_t{1} << maxbits);
}

/// one code-length sequence with its own repeat state (the reference keeps the HLIT and
/// HDIST sequences apart; a run may neither start with 16 nor pass the end)
inline auto read_lengths(huffman::bit_span& bits, const huffman::table<std::uint8_t>& cl, std::uint16_t n)
    -> compat::expected<dyn_table, DecompressStatus> {
  std::vector<std::uint8_t> lens(n, 0);
  for (std::uint16_t i = 0; i < n; ++i) {
    const auto c = huffman::decode_one(cl, bits);
    if (!c.has_value()) return compat::unexpected{DecompressStatus::InvalidLitOrLen};
    bits.consume(c.encoded_size());
    const std::uint8_t s = c.symbol();
    if (s < 16) {
      lens[i] = s;
      continue;
    }
    if (s > 18) return compat::unexpected{DecompressStatus::InvalidLitOrLen};
    const auto x = pop_bits(bits, s == 16 ? 2 : s == 17 ? 3 : 7);
    if (!x) return compat::unexpected{DecompressStatus::Error};
    const auto rep = static_cast<std::uint16_t>(*x + (s == 18 ? 11 : 3));
    if ((s == 16 && i == 0) || i + rep > n) return compat::unexpected{DecompressStatus::Error};
    const std::uint8_t v = s == 16 ? lens[i - 1] : std::uint8_t{0};
    std::fill_n(lens.begin() + i, rep, v);
    i = static_cast<std::uint16_t>(i + rep - 1);
  }
  std::vector<std::pair<huffman::symbol_span<std::uint16_t>, std::uint8_t>> pairs;
  for (std::uint16_t i = 0; i < n; ++i)
    if (lens[i]) pairs.emplace_back(huffman::symbol_span<std::uint16_t>{i}, lens[i]);
  if (oversubscribed(pairs, 15)) return compat::unexpected{DecompressStatus::Error};
  return dyn_table{huffman::symbol_bitsize, pairs};
}

}  // namespace detail

/// Decompresses raw DEFLATE `src` into `dst` (sized by the caller to the exact output length).
/// produced (optional): bytes written -- an extra the reference does not report (src/decompress.hpp:63-64).
inline auto decompress(std::span<const std::byte> src, std::span<std::byte> dst, std::ptrdiff_t* produced)
    -> DecompressStatus {
  huffman::bit_span bits{src};
  std::ptrdiff_t written_here{};
  std::ptrdiff_t& written = produced ? *produced : written_here;
  written = 0;
  for (bool was_final = false; !was_final;) {
    const auto header = detail::read_header(bits);
    if (!header) return header.error();
    was_final = header->final;
    if (header->type == detail::BlockType::NoCompression) {
      bits.consume_to_byte_boundary();
      if (std::ranges::size(bits) < 32) return DecompressStatus::Error;
      const std::uint16_t len = bits.pop_16();
      const std::uint16_t nlen = bits.pop_16();
      if (len != static_cast<std::uint16_t>(~nlen)) return DecompressStatus::NoCompressionLenMismatch;
      if (static_cast<std::size_t>(std::ranges::size(bits)) < std::size_t{len} * CHAR_BIT) return DecompressStatus::SrcTooSmall;
      if (dst.size() - static_cast<std::size_t>(written) < len) return DecompressStatus::DstTooSmall;
      std::copy_n(bits.byte_data(), len, dst.begin() + written);
      bits.consume(std::size_t{len} * CHAR_BIT);
      written += len;
    } else if (header->type == detail::BlockType::FixedHuffman) {
      const auto& [lt, dt] = detail::fixed_tables();
      if (const auto st = detail::inflate_block(bits, dst, written, lt, dt); st != DecompressStatus::Success) return st;
    } else {
      const auto hlit = detail::pop_bits(bits, 5), hdist = detail::pop_bits(bits, 5), hclen = detail::pop_bits(bits, 4);
      if (!hlit || !hdist || !hclen) return DecompressStatus::Error;
      std::vector<std::pair<huffman::symbol_span<std::uint8_t>, std::uint8_t>> clp;
      for (std::size_t i = 0; i < std::size_t{*hclen} + 4; ++i) {
        const auto v = detail::pop_bits(bits, 3);
        if (!v) return DecompressStatus::Error;
        if (*v) clp.emplace_back(huffman::symbol_span<std::uint8_t>{detail::kCodeLengthOrder[i]}, static_cast<std::uint8_t>(*v));
      }
      if (detail::oversubscribed(clp, 7)) return DecompressStatus::Error;
      const huffman::table<std::uint8_t> cl{huffman::symbol_bitsize, clp};
      const auto lt = detail::read_lengths(bits, cl, static_cast<std::uint16_t>(257 + *hlit));
      if (!lt) return lt.error();
      const auto dt = detail::read_lengths(bits, cl, static_cast<std::uint16_t>(1 + *hdist));
      if (!dt) return dt.error();
      if (const auto st = detail::inflate_block(bits, dst, written, *lt, *dt); st != DecompressStatus::Success) return st;
    }
  }
  return DecompressStatus::Success;
}

inline auto decompress(std::span<const std::byte> src, std::span<std::byte> dst) -> DecompressStatus {
  return decompress(src, dst, static_cast<std::ptrdiff_t*>(nullptr));
}

template <std::ranges::contiguous_range R>
  requires std::same_as<std::ranges::range_value_t<R>, std::byte>
auto decompress(const R& src, std::span<std::byte> dst) {
  return decompress(std::span<const std::byte>{src.data(), src.size()}, dst);
}

}  // namespace starflate
