// huffman::bit -- strongly typed bit (API of /root/reference/huffman/src/bit.hpp:14-65).
#pragma once
#include <cassert>
#include <ostream>

namespace starflate::huffman {

class bit {
  bool v_{};

 public:
  bit() = default;
  constexpr explicit bit(int value) : v_{value == 1} { assert(value == 0 || value == 1); }
  constexpr explicit bit(char value) : v_{value == '1'} { assert(value == '0' || value == '1'); }
  constexpr explicit bit(bool value) : v_{value} {}
  constexpr explicit operator bool() const noexcept { return v_; }
  constexpr explicit operator char() const noexcept { return v_ ? '1' : '0'; }
  friend auto operator<<(std::ostream& os, bit b) -> std::ostream& { return os << static_cast<char>(b); }
  friend auto operator==(bit, bit) -> bool = default;
};

namespace detail {
inline void bit_literal_must_be_0_or_1() {}  // not constexpr: reaching it fails constant evaluation
}
namespace literals {
consteval auto operator""_b(unsigned long long n) -> bit {
  if (n > 1) detail::bit_literal_must_be_0_or_1();
  return bit{static_cast<int>(n)};
}
}  // namespace literals

}  // namespace starflate::huffman
