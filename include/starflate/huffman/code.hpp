// huffman::code -- (bitsize, value) with the most significant code bit first
// (API of /root/reference/huffman/src/code.hpp:19-144).
#pragma once
#include "starflate/huffman/bit.hpp"

#include <bit>
#include <cassert>
#include <compare>
#include <cstddef>
#include <cstdint>
#include <ostream>
#include <vector>

namespace starflate::huffman {

class code {
  std::uint8_t bitsize_{};
  std::size_t value_{};

 public:
  code() = default;
  /// @pre value fits in bitsize bits
  constexpr code(std::uint8_t bitsize, std::size_t value) : bitsize_{bitsize}, value_{value} {
    assert(std::size_t{64} - static_cast<std::size_t>(std::countl_zero(value)) <= std::size_t{bitsize});
  }
  [[nodiscard]] constexpr auto bitsize() const -> std::uint8_t { return bitsize_; }
  [[nodiscard]] constexpr auto value() const -> std::size_t { return value_; }

  /// the code as bits, left (most significant) to right
  [[nodiscard]] auto bit_view() const -> std::vector<bit> {
    std::vector<bit> out;
    for (std::size_t n = bitsize_; n-- > 0;) out.emplace_back(bool((value_ >> n) & 1U));
    return out;
  }

  /// left pad c with b
  friend constexpr auto operator>>(bit b, code& c) -> code& {
    if (b) c.value_ += std::size_t{1} << c.bitsize_;
    ++c.bitsize_;
    return c;
  }
  friend constexpr auto operator>>(bit b, code&& c) -> code&& {
    b >> c;
    return static_cast<code&&>(c);
  }
  /// right pad c with b
  friend constexpr auto operator<<(code& c, bit b) -> code& {
    c.value_ = (c.value_ << 1U) | static_cast<std::size_t>(bool(b));
    ++c.bitsize_;
    return c;
  }
  friend constexpr auto operator<<(code&& c, bit b) -> code&& {
    c << b;
    return static_cast<code&&>(c);
  }
  friend auto operator<<(std::ostream& os, const code& c) -> std::ostream& {
    for (std::size_t n = c.bitsize_; n-- > 0;) os << (((c.value_ >> n) & 1U) ? '1' : '0');
    return os;
  }
  [[nodiscard]] friend auto operator<=>(const code&, const code&) = default;
};

namespace detail {
inline void code_literal_must_be_binary() {}  // not constexpr: reaching it fails constant evaluation
}
namespace literals {
/// 110_c == code{3, 0b110}
template <char... Bits>
consteval auto operator""_c() -> code {
  std::size_t v = 0;
  for (char c : {Bits...}) {
    if (c != '0' && c != '1') detail::code_literal_must_be_binary();
    v = (v << 1U) | static_cast<std::size_t>(c == '1');
  }
  return {static_cast<std::uint8_t>(sizeof...(Bits)), v};
}
}  // namespace literals

}  // namespace starflate::huffman
