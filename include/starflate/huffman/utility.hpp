// Tags, byte_array and the symbol concept (API of /root/reference/huffman/src/utility.hpp:10-41).
#pragma once
#include <array>
#include <concepts>
#include <cstddef>

namespace starflate::huffman {

template <class T, std::size_t N>
using c_array = T[N];

struct table_contents_tag {
  explicit table_contents_tag() = default;
};
inline constexpr table_contents_tag table_contents{};

struct symbol_bitsize_tag {
  explicit symbol_bitsize_tag() = default;
};
inline constexpr symbol_bitsize_tag symbol_bitsize{};

template <class... Ts>
constexpr auto byte_array(Ts... values) {
  return std::array<std::byte, sizeof...(Ts)>{std::byte(values)...};
}

template <class T>
concept symbol = std::regular<T> && std::totally_ordered<T>;

}  // namespace starflate::huffman
