// huffman::encoding<S> -- a symbol with its code (API of /root/reference/huffman/src/encoding.hpp:16-49).
#pragma once
#include "starflate/huffman/code.hpp"
#include "starflate/huffman/utility.hpp"

#include <ostream>

namespace starflate::huffman {

template <symbol Symbol>
struct encoding : code {
  using symbol_type = Symbol;
  symbol_type symbol{};

  encoding() = default;
  constexpr explicit encoding(symbol_type s) : symbol{s} {}
  constexpr explicit encoding(symbol_type s, code c) : code{c}, symbol{s} {}

  friend auto operator<<(std::ostream& os, const encoding& e) -> std::ostream& {
    return os << +e.bitsize() << '\t' << static_cast<const code&>(e) << '\t' << e.value() << "\t`" << e.symbol << '`';
  }
  [[nodiscard]] friend auto operator<=>(const encoding&, const encoding&) = default;
};

}  // namespace starflate::huffman
