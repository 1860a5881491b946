// huffman::decode / decode_one (API of /root/reference/huffman/src/decode.hpp:25-102):
// code bits are consumed most-significant first, one table::find per bit.
#pragma once
#include "starflate/huffman/bit_span.hpp"
#include "starflate/huffman/code.hpp"
#include "starflate/huffman/table.hpp"

#include <cassert>
#include <cstdint>
#include <iterator>

namespace starflate::huffman {

template <symbol Symbol>
class decode_result {
 public:
  static constexpr std::uint8_t kInvalidEncodedSize = 0;
  constexpr decode_result(Symbol s, std::uint8_t n) : symbol_{s}, encoded_size_{n} {}
  [[nodiscard]] constexpr auto has_value() const -> bool { return encoded_size_ != kInvalidEncodedSize; }
  [[nodiscard]] constexpr auto symbol() const -> Symbol { return symbol_; }
  [[nodiscard]] constexpr auto encoded_size() const -> std::uint8_t { return encoded_size_; }

 private:
  Symbol symbol_;
  std::uint8_t encoded_size_;
};

/// one symbol from the front of bits (bits is not consumed); encoded_size()==0 when no code matches
template <symbol Symbol, std::size_t Extent>
constexpr auto decode_one(const table<Symbol, Extent>& code_table, bit_span bits) -> decode_result<Symbol> {
  code current{};
  auto pos = code_table.begin();
  for (auto b : bits) {
    current << b;
    const auto found = code_table.find(current, pos);
    if (found) return {(*found)->symbol, (*found)->bitsize()};
    if (found.error() == code_table.end()) break;
    pos = found.error();
  }
  return {Symbol{}, decode_result<Symbol>::kInvalidEncodedSize};
}

/// symbols until the bits run out or a code is not in the table
template <symbol Symbol, std::size_t Extent, std::output_iterator<Symbol> O>
constexpr auto decode(const table<Symbol, Extent>& code_table, bit_span bits, O output) -> O {
  while (!bits.empty()) {
    const auto r = decode_one(code_table, bits);
    if (!r.has_value()) break;
    *output = r.symbol();
    ++output;
    bits.consume(r.encoded_size());
  }
  return output;
}

}  // namespace starflate::huffman
