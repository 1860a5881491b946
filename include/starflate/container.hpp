// zlib (RFC 1950) and gzip (RFC 1952) wrappers around a raw DEFLATE stream -- the bytes the reference's
// fixture tool strips (/root/reference/tools/deflate_compress.py:8-13).  Host-side: the checksums, and a
// decompress() overload that validates the wrapper, inflates the body with the raw decompress()
// (/root/reference/src/decompress.hpp:63-71 signature) and verifies the trailer.  The GPU side writes the
// same wrappers from compress() (compress_options::container).
#pragma once
#include "starflate/decompress.hpp"

#include <array>
#include <cstddef>
#include <cstdint>
#include <span>

namespace starflate {

enum class Container : std::uint8_t {
  Raw,   // RFC 1951 only (what the reference reads)
  Zlib,  // 78 9C .. Adler-32 (big-endian)
  Gzip,  // 1F 8B 08 .. CRC-32, ISIZE (little-endian)
};

namespace detail {
inline constexpr auto kCrcTable = [] {
  std::array<std::uint32_t, 256> t{};
  for (std::uint32_t i = 0; i < 256; ++i) {
    std::uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1U) != 0U ? (c >> 1U) ^ 0xEDB88320U : c >> 1U;
    t[i] = c;
  }
  return t;
}();
inline auto u8(std::byte b) -> std::uint32_t { return std::to_integer<std::uint32_t>(b); }
}  // namespace detail

/// RFC 1952 section 8
constexpr auto crc32(std::span<const std::byte> data) -> std::uint32_t {
  std::uint32_t c = 0xFFFFFFFFU;
  for (const auto b : data) c = detail::kCrcTable[(c ^ std::to_integer<std::uint32_t>(b)) & 0xFFU] ^ (c >> 8U);
  return ~c;
}

/// RFC 1950 section 8.2
constexpr auto adler32(std::span<const std::byte> data) -> std::uint32_t {
  constexpr std::uint32_t mod = 65521;
  std::uint32_t a = 1;
  std::uint32_t b = 0;
  std::size_t i = 0;
  while (i < data.size()) {
    const std::size_t stop = std::min(data.size(), i + 5552);  // largest run that cannot overflow 32 bits
    for (; i < stop; ++i) {
      a += std::to_integer<std::uint32_t>(data[i]);
      b += a;
    }
    a %= mod;
    b %= mod;
  }
  return (b << 16U) | a;
}

/// decompress() for a wrapped stream.  `src` is exactly the wrapped stream.  Zlib: `dst` is exactly the
/// output size (the format does not carry it); Gzip: dst.size() >= ISIZE.  A malformed wrapper or a
/// checksum / ISIZE mismatch is DecompressStatus::Error; everything else is the raw decoder's status.
inline auto decompress(std::span<const std::byte> src, std::span<std::byte> dst, Container container) -> DecompressStatus {
  using detail::u8;
  switch (container) {
    case Container::Raw: return decompress(src, dst);
    case Container::Zlib: {
      if (src.size() < 6) return DecompressStatus::SrcTooSmall;
      const std::uint32_t cmf = u8(src[0]);
      const std::uint32_t flg = u8(src[1]);
      if ((cmf & 0x0FU) != 8 || (cmf >> 4U) > 7 || ((cmf << 8U) | flg) % 31 != 0 || (flg & 0x20U) != 0) return DecompressStatus::Error;
      const auto st = decompress(src.subspan(2, src.size() - 6), dst);
      if (st != DecompressStatus::Success) return st;
      const auto tr = src.last(4);
      const std::uint32_t want = (u8(tr[0]) << 24U) | (u8(tr[1]) << 16U) | (u8(tr[2]) << 8U) | u8(tr[3]);
      return adler32(dst) == want ? DecompressStatus::Success : DecompressStatus::Error;
    }
    case Container::Gzip: {
      if (src.size() < 18) return DecompressStatus::SrcTooSmall;
      if (u8(src[0]) != 0x1F || u8(src[1]) != 0x8B || u8(src[2]) != 8) return DecompressStatus::Error;
      const std::uint32_t flg = u8(src[3]);
      if ((flg & 0xE0U) != 0) return DecompressStatus::Error;  // reserved bits
      std::size_t p = 10;
      const std::size_t end = src.size() - 8;
      if ((flg & 0x04U) != 0) {  // FEXTRA
        if (p + 2 > end) return DecompressStatus::SrcTooSmall;
        p += 2 + (u8(src[p]) | (u8(src[p + 1]) << 8U));
      }
      for (const std::uint32_t bit : {0x08U, 0x10U}) {  // FNAME, FCOMMENT: zero-terminated
        if ((flg & bit) == 0) continue;
        while (p < end && u8(src[p]) != 0) ++p;
        ++p;
      }
      if ((flg & 0x02U) != 0) p += 2;  // FHCRC
      if (p > end) return DecompressStatus::SrcTooSmall;
      const auto tr = src.last(8);
      const auto le32 = [&](std::size_t k) { return u8(tr[k]) | (u8(tr[k + 1]) << 8U) | (u8(tr[k + 2]) << 16U) | (u8(tr[k + 3]) << 24U); };
      const std::uint32_t isize = le32(4);
      if (dst.size() < isize) return DecompressStatus::DstTooSmall;
      // the body must produce exactly ISIZE bytes: a longer one runs into DstTooSmall here, never past dst.first(isize)
      std::ptrdiff_t written = 0;
      const auto st = decompress(src.subspan(p, end - p), dst.first(isize), &written);
      if (st != DecompressStatus::Success) return st;
      if (static_cast<std::uint64_t>(written) != isize) return DecompressStatus::Error;
      return crc32(dst.first(isize)) == le32(0) ? DecompressStatus::Success : DecompressStatus::Error;
    }
  }
  return DecompressStatus::Error;
}

template <std::ranges::contiguous_range R>
  requires std::same_as<std::ranges::range_value_t<R>, std::byte>
auto decompress(const R& src, std::span<std::byte> dst, Container container) {
  return decompress(std::span<const std::byte>{src.data(), src.size()}, dst, container);
}

}  // namespace starflate
