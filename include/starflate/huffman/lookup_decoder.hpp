// huffman::lookup_decoder -- decode_one() with one array read instead of one table::find per bit
// (the reference's per-bit walk is /root/reference/huffman/src/decode.hpp:83-102).
// Built from a table; codes longer than `Bits`, and inputs shorter than the code, fall back to the walk, so
// the result is always the one huffman::decode_one(table, bits) gives.
#pragma once
#include "starflate/huffman/bit_span.hpp"
#include "starflate/huffman/decode.hpp"
#include "starflate/huffman/table.hpp"

#include <array>
#include <cstddef>
#include <cstdint>
#include <ranges>

namespace starflate::huffman {

template <symbol Symbol, std::uint8_t Bits = 9>
class lookup_decoder {
  static_assert(Bits >= 1 && Bits <= 15);
  struct entry {
    Symbol symbol{};
    std::uint8_t bitsize{};  // 0: no code of at most Bits bits starts with these bits
  };
  std::array<entry, (std::size_t{1} << Bits)> lut_{};

 public:
  template <std::size_t Extent>
  constexpr explicit lookup_decoder(const table<Symbol, Extent>& t) {
    for (const auto& e : t) {  // ascending code length: a shorter code keeps its slots, as the per-bit walk finds it first
      const std::uint8_t n = e.bitsize();
      if (n == 0 || n > Bits) continue;
      std::size_t rev = 0;  // the stream carries code bits most significant first
      for (std::uint8_t k = 0; k < n; ++k) rev |= ((e.value() >> k) & 1U) << (n - 1U - k);
      for (std::size_t i = rev; i < lut_.size(); i += std::size_t{1} << n)
        if (lut_[i].bitsize == 0) lut_[i] = entry{e.symbol, n};
    }
  }

  /// same result as huffman::decode_one(t, bits); `t` must be the table this decoder was built from
  template <std::size_t Extent>
  constexpr auto decode_one(const table<Symbol, Extent>& t, bit_span bits) const -> decode_result<Symbol> {
    const entry& e = lut_[bits.peek(Bits)];
    if (e.bitsize != 0 && e.bitsize <= std::ranges::size(bits)) return {e.symbol, e.bitsize};
    return huffman::decode_one(t, bits);
  }
};

template <symbol Symbol, std::size_t Extent>
lookup_decoder(const table<Symbol, Extent>&) -> lookup_decoder<Symbol>;

}  // namespace starflate::huffman
