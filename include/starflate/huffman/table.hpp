// huffman::table<Symbol, Extent> -- canonical ("DEFLATE form") prefix-code table.
// Public surface of /root/reference/huffman/src/table.hpp:74-528: four constructors
// (frequencies, data, explicit contents, symbol->bitsize), begin()/end() over
// encoding<Symbol> sorted by (bitsize, symbol), find(code[, pos]) and operator<<.
// Own implementation: a flat array of entries plus the distance to the next bitsize group.
#pragma once
#include "starflate/compat/expected.hpp"
#include "starflate/huffman/code.hpp"
#include "starflate/huffman/encoding.hpp"
#include "starflate/huffman/symbol_span.hpp"
#include "starflate/huffman/utility.hpp"

#include <algorithm>
#include <array>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iterator>
#include <optional>
#include <ostream>
#include <ranges>
#include <span>
#include <tuple>
#include <utility>
#include <vector>

namespace starflate::huffman {
namespace detail {

template <class T, std::size_t N>
class fixed_vector {
  std::array<T, N> a_{};
  std::size_t n_{};

 public:
  constexpr auto push_back(const T& v) -> void { assert(n_ < N); a_[n_++] = v; }
  constexpr auto size() const -> std::size_t { return n_; }
  constexpr auto data() -> T* { return a_.data(); }
  constexpr auto data() const -> const T* { return a_.data(); }
  constexpr auto begin() -> T* { return a_.data(); }
  constexpr auto end() -> T* { return a_.data() + n_; }
  constexpr auto begin() const -> const T* { return a_.data(); }
  constexpr auto end() const -> const T* { return a_.data() + n_; }
  constexpr auto operator[](std::size_t i) -> T& { return a_[i]; }
  constexpr auto operator[](std::size_t i) const -> const T& { return a_[i]; }
  constexpr auto reserve(std::size_t) -> void {}
};

template <class T>
struct is_symbol_span : std::false_type {};
template <class S>
struct is_symbol_span<symbol_span<S>> : std::true_type {};

template <class T>
concept pair_like = requires { typename std::tuple_size<std::remove_cvref_t<T>>::type; } &&
                    (std::tuple_size_v<std::remove_cvref_t<T>> == 2);

template <class R>
constexpr auto static_extent_of() -> std::size_t {
  using U = std::remove_cvref_t<R>;
  if constexpr (std::is_bounded_array_v<U>) return std::extent_v<U>;
  else if constexpr (requires { std::tuple_size<U>::value; }) return std::tuple_size_v<U>;
  else return std::dynamic_extent;
}

}  // namespace detail

template <symbol Symbol, std::size_t Extent = std::dynamic_extent>
class table {
 public:
  using symbol_type = Symbol;
  using encoding_type = encoding<Symbol>;

 private:
  struct node : encoding_type {
    std::size_t skip{1};  // entries from this one to the first entry of the next bitsize group
    std::size_t weight{};
  };
  using storage_type =
      std::conditional_t<Extent == std::dynamic_extent, std::vector<node>, detail::fixed_vector<node, Extent>>;
  storage_type nodes_{};

  constexpr auto set_skips() -> void {
    std::size_t run = 0;
    for (std::size_t i = nodes_.size(); i-- > 0;) {
      run = (i + 1 < nodes_.size() && nodes_[i + 1].bitsize() == nodes_[i].bitsize()) ? run + 1 : 1;
      nodes_[i].skip = run;
    }
  }

  // RFC 1951 3.2.2: order by (bitsize, symbol), consecutive values within a bitsize,
  // first value of a longer bitsize = (last value + 1) << (bitsize difference)
  constexpr auto canonicalize() -> void {
    std::sort(nodes_.begin(), nodes_.end(), [](const node& x, const node& y) {
      return x.bitsize() != y.bitsize() ? x.bitsize() < y.bitsize() : std::less<>{}(x.symbol, y.symbol);
    });
    std::size_t next = 0;
    std::uint8_t len = 0;
    for (auto& e : nodes_) {
      next <<= (e.bitsize() - len);
      len = e.bitsize();
      static_cast<code&>(e) = code{len, next};
      ++next;
    }
    set_skips();
  }

  // unrestricted Huffman code lengths: two-queue merge over entries sorted ascending by
  // (weight, symbol); among equal weights existing nodes go before newly merged ones
  constexpr auto assign_huffman_bitsizes() -> void {
    const std::size_t m = nodes_.size();
    if (m == 0) return;
    if (m == 1) {
      static_cast<code&>(nodes_[0]) = code{1, 0};
      return;
    }
    std::sort(nodes_.begin(), nodes_.end(), [](const node& x, const node& y) {
      return x.weight != y.weight ? x.weight < y.weight : std::less<>{}(x.symbol, y.symbol);
    });
    assert(std::adjacent_find(nodes_.begin(), nodes_.end(),
                              [](const node& x, const node& y) { return x.symbol == y.symbol; }) == nodes_.end() &&
           "a `table` cannot contain duplicate symbols");
    std::vector<std::size_t> w(2 * m - 1), parent(2 * m - 1, 0);
    for (std::size_t i = 0; i < m; ++i) w[i] = nodes_[i].weight;
    std::size_t i = 0, j = m, k = m;
    auto take = [&]() -> std::size_t { return (i < m && (j >= k || w[i] <= w[j])) ? i++ : j++; };
    while (k < 2 * m - 1) {
      const std::size_t a = take(), b = take();
      w[k] = w[a] + w[b];
      parent[a] = parent[b] = k;
      ++k;
    }
    std::vector<std::uint8_t> depth(2 * m - 1, 0);
    for (std::size_t v = 2 * m - 2; v-- > 0;) depth[v] = static_cast<std::uint8_t>(depth[parent[v]] + 1);
    for (std::size_t q = 0; q < m; ++q) static_cast<code&>(nodes_[q]) = code{depth[q], 0};
  }

  template <class R>
  constexpr auto load_frequencies(const R& frequencies, std::optional<symbol_type> eot) -> void {
    nodes_.reserve(std::ranges::size(frequencies) + 1);
    if (eot) {
      node e{};
      e.symbol = *eot;
      e.weight = 1;
      nodes_.push_back(e);
    }
    for (const auto& [s, f] : frequencies) {
      assert(f > 0);
      node e{};
      e.symbol = static_cast<symbol_type>(s);
      e.weight = static_cast<std::size_t>(f);
      nodes_.push_back(e);
    }
  }

 public:
  /// random-access iterator over encoding<Symbol>
  class const_iterator {
    const node* p_{nullptr};

   public:
    using difference_type = std::ptrdiff_t;
    using value_type = encoding_type;
    using reference = const encoding_type&;
    using pointer = const encoding_type*;
    using iterator_category = std::random_access_iterator_tag;
    using iterator_concept = std::random_access_iterator_tag;

    const_iterator() = default;
    constexpr explicit const_iterator(const node* p) : p_{p} {}
    constexpr auto base() const -> const node* { return p_; }
    constexpr auto operator*() const -> reference { return *p_; }
    constexpr auto operator->() const -> pointer { return p_; }
    constexpr auto operator[](difference_type n) const -> reference { return p_[n]; }
    constexpr auto operator+=(difference_type n) -> const_iterator& { p_ += n; return *this; }
    constexpr auto operator-=(difference_type n) -> const_iterator& { p_ -= n; return *this; }
    constexpr auto operator++() -> const_iterator& { ++p_; return *this; }
    constexpr auto operator++(int) -> const_iterator { auto t = *this; ++p_; return t; }
    constexpr auto operator--() -> const_iterator& { --p_; return *this; }
    constexpr auto operator--(int) -> const_iterator { auto t = *this; --p_; return t; }
    friend constexpr auto operator+(const_iterator i, difference_type n) -> const_iterator { return i += n; }
    friend constexpr auto operator+(difference_type n, const_iterator i) -> const_iterator { return i += n; }
    friend constexpr auto operator-(const_iterator i, difference_type n) -> const_iterator { return i -= n; }
    friend constexpr auto operator-(const_iterator a, const_iterator b) -> difference_type { return a.p_ - b.p_; }
    friend constexpr auto operator==(const_iterator a, const_iterator b) -> bool { return a.p_ == b.p_; }
    friend constexpr auto operator<=>(const_iterator a, const_iterator b) { return a.p_ <=> b.p_; }
  };

  table() = default;

  /// from a symbol -> frequency mapping (+ optional end-of-transmission symbol of frequency 1)
  template <std::ranges::sized_range R>
    requires detail::pair_like<std::ranges::range_value_t<R>> &&
             std::convertible_to<std::tuple_element_t<0, std::ranges::range_value_t<R>>, symbol_type> &&
             std::integral<std::tuple_element_t<1, std::ranges::range_value_t<R>>>
  constexpr table(const R& frequencies, std::optional<symbol_type> eot) {
    load_frequencies(frequencies, eot);
    assign_huffman_bitsizes();
    canonicalize();
  }
  template <std::ranges::sized_range R>
    requires detail::pair_like<std::ranges::range_value_t<R>> &&
             std::convertible_to<std::tuple_element_t<0, std::ranges::range_value_t<R>>, symbol_type> &&
             std::integral<std::tuple_element_t<1, std::ranges::range_value_t<R>>>
  constexpr explicit table(const R& frequencies) : table{frequencies, std::nullopt} {}

  /// from a sequence of symbols (+ optional eot)
  template <std::ranges::input_range R>
    requires std::convertible_to<std::ranges::range_reference_t<R>, symbol_type> &&
             (!detail::pair_like<std::ranges::range_value_t<R>>)
  constexpr explicit table(const R& data, std::optional<symbol_type> eot) {
    std::vector<std::pair<symbol_type, std::size_t>> freq;
    for (const auto& s : data) {
      auto it = std::find_if(freq.begin(), freq.end(), [&](const auto& p) { return p.first == s; });
      if (it == freq.end()) freq.emplace_back(static_cast<symbol_type>(s), std::size_t{1});
      else ++it->second;
    }
    load_frequencies(freq, eot);
    assign_huffman_bitsizes();
    canonicalize();
  }
  template <std::ranges::input_range R>
    requires std::convertible_to<std::ranges::range_reference_t<R>, symbol_type> &&
             (!detail::pair_like<std::ranges::range_value_t<R>>)
  constexpr explicit table(const R& data) : table{data, std::nullopt} {}

  /// explicit contents, (code, symbol) pairs already in DEFLATE canonical order
  template <std::ranges::sized_range R>
    requires detail::pair_like<std::ranges::range_value_t<R>> &&
             std::same_as<std::remove_cvref_t<std::tuple_element_t<0, std::ranges::range_value_t<R>>>, code>
  constexpr table(table_contents_tag, const R& map) {
    nodes_.reserve(std::ranges::size(map));
    for (const auto& [c, s] : map) {
      node e{};
      static_cast<code&>(e) = c;
      e.symbol = static_cast<symbol_type>(s);
      nodes_.push_back(e);
    }
    assert(std::is_sorted(nodes_.begin(), nodes_.end(), [](const node& x, const node& y) {
             return x.bitsize() != y.bitsize() ? x.bitsize() < y.bitsize() : x.value() < y.value();
           }) && "table contents are not provided in DEFLATE canonical form");
    set_skips();
  }
  template <std::size_t N>
  constexpr table(table_contents_tag, const c_array<std::pair<code, symbol_type>, N>& map)
      : table{table_contents, std::span<const std::pair<code, symbol_type>, N>{map}} {}

  /// from symbol (or inclusive symbol span) -> bitsize
  template <std::ranges::input_range R>
    requires detail::pair_like<std::ranges::range_value_t<R>> &&
             std::convertible_to<std::tuple_element_t<0, std::ranges::range_value_t<R>>, symbol_span<symbol_type>>
  constexpr table(symbol_bitsize_tag, const R& map) {
    for (const auto& [ss, bits] : map) {
      const symbol_span<symbol_type> span = ss;
      symbol_type s = span.first();
      for (std::size_t c = span.count(); c-- > 0; ++s) {
        node e{};
        static_cast<code&>(e) = code{static_cast<std::uint8_t>(bits), 0};
        e.symbol = s;
        nodes_.push_back(e);
      }
    }
    canonicalize();
  }
  template <std::size_t N>
  constexpr table(symbol_bitsize_tag, const c_array<std::pair<symbol_span<symbol_type>, std::uint8_t>, N>& map)
      : table{symbol_bitsize, std::span<const std::pair<symbol_span<symbol_type>, std::uint8_t>, N>{map}} {}

  [[nodiscard]] constexpr auto begin() const -> const_iterator { return const_iterator{nodes_.data()}; }
  [[nodiscard]] constexpr auto end() const -> const_iterator { return const_iterator{nodes_.data() + nodes_.size()}; }
  [[nodiscard]] constexpr auto size() const -> std::size_t { return nodes_.size(); }

  /// Finds the entry whose code equals c, searching bitsize groups from pos on.
  /// Value: iterator to the entry.  Error: iterator to the first entry with a longer
  /// bitsize than c (keep feeding bits), or end() (c cannot be completed in this table).
  [[nodiscard]] constexpr auto find(code c) const -> compat::expected<const_iterator, const_iterator> {
    return find(c, begin());
  }
  [[nodiscard]] constexpr auto find(code c, const_iterator pos) const
      -> compat::expected<const_iterator, const_iterator> {
    using R = compat::expected<const_iterator, const_iterator>;
    while (pos != end()) {
      if (pos->bitsize() > c.bitsize()) break;
      const std::size_t group = pos.base()->skip;
      if (pos->bitsize() == c.bitsize()) {
        const std::size_t d = c.value() - pos->value();  // unsigned: a smaller c.value() wraps and misses
        if (d < group) return R{std::in_place, pos + static_cast<std::ptrdiff_t>(d)};
      }
      pos += static_cast<std::ptrdiff_t>(group);
    }
    return R{compat::unexpect, pos};
  }

  friend auto operator<<(std::ostream& os, const table& t) -> std::ostream& {
    os << "Bits\tCode\tValue\tSymbol\n";
    for (const auto& e : t) os << e << '\n';
    return os;
  }
};

/// north_star spells the type `huffman::code_table`
template <symbol Symbol, std::size_t Extent = std::dynamic_extent>
using code_table = table<Symbol, Extent>;

// deduction guides
template <class R>
  requires detail::pair_like<std::ranges::range_value_t<R>> &&
           std::integral<std::tuple_element_t<1, std::ranges::range_value_t<R>>>
table(const R&) -> table<std::remove_cvref_t<std::tuple_element_t<0, std::ranges::range_value_t<R>>>,
                        detail::static_extent_of<R>()>;
template <class R, class S>
  requires detail::pair_like<std::ranges::range_value_t<R>> &&
           std::integral<std::tuple_element_t<1, std::ranges::range_value_t<R>>>
table(const R&, S) -> table<S, (detail::static_extent_of<R>() == std::dynamic_extent ? std::dynamic_extent
                                                                                    : detail::static_extent_of<R>() + 1)>;
template <class R>
  requires(!detail::pair_like<std::ranges::range_value_t<R>>)
table(const R&) -> table<std::ranges::range_value_t<R>>;
template <class R, class S>
  requires(!detail::pair_like<std::ranges::range_value_t<R>>)
table(const R&, S) -> table<S>;
template <class S, std::size_t N>
table(table_contents_tag, const c_array<std::pair<code, S>, N>&) -> table<S, N>;
template <class R>
  requires detail::pair_like<std::ranges::range_value_t<R>>
table(table_contents_tag, const R&)
    -> table<std::remove_cvref_t<std::tuple_element_t<1, std::ranges::range_value_t<R>>>, detail::static_extent_of<R>()>;
template <class S, class I, std::size_t N>
  requires(!detail::is_symbol_span<S>::value)
table(symbol_bitsize_tag, const c_array<std::pair<S, I>, N>&) -> table<S, N>;

}  // namespace starflate::huffman
