// huffman::symbol_span<S> -- inclusive range of symbols (API of /root/reference/huffman/src/symbol_span.hpp:12-55).
#pragma once
#include "starflate/huffman/utility.hpp"

#include <cassert>
#include <ranges>

namespace starflate::huffman {

template <symbol S>
class symbol_span : public std::ranges::view_interface<symbol_span<S>> {
  S first_;
  S last_;
  using range_type = std::ranges::iota_view<S, S>;
  constexpr auto rng() const -> range_type { return range_type{first_, static_cast<S>(last_ + S{1})}; }

 public:
  using symbol_type = S;
  using iterator = std::ranges::iterator_t<range_type>;
  constexpr symbol_span(symbol_type s) : symbol_span{s, s} {}
  /// @pre first <= last
  constexpr symbol_span(symbol_type first, symbol_type last) : first_{first}, last_{last} { assert(first <= last); }
  [[nodiscard]] constexpr auto begin() const -> iterator { return rng().begin(); }
  [[nodiscard]] constexpr auto end() const -> iterator { return rng().end(); }
  [[nodiscard]] constexpr auto count() const -> std::size_t { return static_cast<std::size_t>(last_ - first_) + 1; }
  [[nodiscard]] constexpr auto first() const -> S { return first_; }
};

}  // namespace starflate::huffman
