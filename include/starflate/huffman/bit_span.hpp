// huffman::bit_span -- non-owning LSB-first view of bits: stream bit i is
// (byte[i / 8] >> (i % 8)) & 1  (API of /root/reference/huffman/src/bit_span.hpp:18-183).
#pragma once
#include "starflate/huffman/bit.hpp"

#include <bit>
#include <cassert>
#include <climits>
#include <compare>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <ranges>

namespace starflate::huffman {

class bit_span : public std::ranges::view_interface<bit_span> {
  const std::byte* data_{nullptr};
  std::size_t bit_size_{};
  std::uint8_t bit_offset_{};  // < CHAR_BIT

 public:
  class iterator {
    const std::byte* base_{nullptr};
    std::size_t off_{};

   public:
    using difference_type = std::ptrdiff_t;
    using iterator_category = std::random_access_iterator_tag;
    using iterator_concept = std::random_access_iterator_tag;
    using value_type = bit;
    using reference = bit;
    using pointer = void;

    iterator() = default;
    constexpr iterator(const std::byte* base, std::size_t off) : base_{base}, off_{off} {}
    constexpr auto operator*() const -> bit {
      return bit{((static_cast<unsigned>(base_[off_ / CHAR_BIT]) >> (off_ % CHAR_BIT)) & 1U) != 0U};
    }
    constexpr auto operator[](difference_type n) const -> bit { return *(*this + n); }
    constexpr auto operator+=(difference_type n) -> iterator& {
      off_ = static_cast<std::size_t>(static_cast<difference_type>(off_) + n);
      return *this;
    }
    constexpr auto operator-=(difference_type n) -> iterator& { return *this += -n; }
    constexpr auto operator++() -> iterator& { return *this += 1; }
    constexpr auto operator++(int) -> iterator { auto t = *this; ++*this; return t; }
    constexpr auto operator--() -> iterator& { return *this -= 1; }
    constexpr auto operator--(int) -> iterator { auto t = *this; --*this; return t; }
    friend constexpr auto operator+(iterator i, difference_type n) -> iterator { return i += n; }
    friend constexpr auto operator+(difference_type n, iterator i) -> iterator { return i += n; }
    friend constexpr auto operator-(iterator i, difference_type n) -> iterator { return i -= n; }
    friend constexpr auto operator-(const iterator& a, const iterator& b) -> difference_type {
      assert(a.base_ == b.base_);
      return static_cast<difference_type>(a.off_) - static_cast<difference_type>(b.off_);
    }
    friend constexpr auto operator==(const iterator& a, const iterator& b) -> bool { return a.off_ == b.off_; }
    friend constexpr auto operator<=>(const iterator& a, const iterator& b) { return a.off_ <=> b.off_; }
  };

  bit_span() = default;
  /// @pre bit_offset < CHAR_BIT
  constexpr bit_span(const std::byte* data, std::size_t bit_size, std::uint8_t bit_offset = {})
      : data_{data}, bit_size_{bit_size}, bit_offset_{bit_offset} {
    assert(bit_offset < CHAR_BIT);
  }
  template <std::ranges::contiguous_range R>
    requires std::ranges::borrowed_range<R> && std::same_as<std::remove_cv_t<std::ranges::range_value_t<R>>, std::byte>
  constexpr bit_span(R&& r) : bit_span(std::ranges::data(r), std::ranges::size(r) * CHAR_BIT) {}

  [[nodiscard]] constexpr auto begin() const -> iterator { return {data_, bit_offset_}; }
  [[nodiscard]] constexpr auto end() const -> iterator { return {data_, bit_offset_ + bit_size_}; }

  /// little-endian integer from a byte-aligned span; @pre aligned and enough bits
  template <std::integral T>
  constexpr auto pop() -> T {
    assert(bit_size_ >= sizeof(T) * CHAR_BIT && bit_offset_ == 0);
    std::make_unsigned_t<T> r{};
    for (std::size_t k = 0; k < sizeof(T); ++k)
      r = static_cast<std::make_unsigned_t<T>>(r | (static_cast<std::make_unsigned_t<T>>(data_[k]) << (CHAR_BIT * k)));
    data_ += sizeof(T);
    bit_size_ -= sizeof(T) * CHAR_BIT;
    return static_cast<T>(r);
  }
  constexpr auto pop_8() -> std::uint8_t { return pop<std::uint8_t>(); }
  constexpr auto pop_16() -> std::uint16_t { return pop<std::uint16_t>(); }

  /// @pre n <= size()
  constexpr auto consume(std::size_t n) & -> bit_span& {
    assert(n <= bit_size_);
    bit_size_ -= n;
    const std::size_t d = bit_offset_ + n;
    data_ += d / CHAR_BIT;
    bit_offset_ = static_cast<std::uint8_t>(d % CHAR_BIT);
    return *this;
  }
  constexpr auto consume(std::size_t n) && -> bit_span&& {
    consume(n);
    return static_cast<bit_span&&>(*this);
  }
  constexpr auto consume_to_byte_boundary() -> void {
    if (bit_offset_) consume(static_cast<std::size_t>(CHAR_BIT - bit_offset_));
  }
  /// The first min(n, size()) bits as an integer, first bit in bit 0 (n <= 24).  Not in the reference's
  /// bit_span: it feeds huffman::lookup_decoder, which replaces one table::find per bit by one array read.
  [[nodiscard]] constexpr auto peek(std::uint8_t n) const -> std::uint32_t {
    assert(n <= 24);
    const std::size_t take = bit_size_ < n ? bit_size_ : n;
    if (take == 0) return 0;
    const std::size_t nbytes = (bit_offset_ + take + CHAR_BIT - 1) / CHAR_BIT;  // <= 4
    std::uint32_t v = 0;
    for (std::size_t k = 0; k < nbytes; ++k) v |= std::to_integer<std::uint32_t>(data_[k]) << (CHAR_BIT * k);
    return (v >> bit_offset_) & ((std::uint32_t{1} << take) - 1U);
  }
  /// @pre byte aligned
  [[nodiscard]] constexpr auto byte_data() const -> const std::byte* {
    assert(bit_offset_ == 0);
    return data_;
  }
};

}  // namespace starflate::huffman

template <>
inline constexpr bool std::ranges::enable_borrowed_range<starflate::huffman::bit_span> = true;
