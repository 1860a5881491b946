// starflate::compat::expected -- the subset of C++23 std::expected the API needs.
// The target image ships libstdc++-11 (no <expected>); when the standard header exists
// the std types are used instead.  (/root/reference uses std::expected directly:
// src/decompress.hpp:40-41, huffman/src/table.hpp:420-423.)
#pragma once
#include <version>
#if __has_include(<expected>) && defined(__cpp_lib_expected)
#include <expected>
namespace starflate::compat {
using std::expected;
using std::unexpect;
using std::unexpect_t;
using std::unexpected;
}  // namespace starflate::compat
#else
#include <cassert>
#include <type_traits>
#include <utility>
#include <variant>
namespace starflate::compat {

template <class E>
class unexpected {
  E e_;

 public:
  constexpr explicit unexpected(E e) : e_{std::move(e)} {}
  constexpr auto error() const& -> const E& { return e_; }
  constexpr auto error() & -> E& { return e_; }
};
template <class E>
unexpected(E) -> unexpected<E>;

struct unexpect_t {
  explicit unexpect_t() = default;
};
inline constexpr unexpect_t unexpect{};

template <class T, class E>
class expected {
  // index 0 = value, 1 = error (T and E may be the same type, as in table::find)
  std::variant<T, E> v_;

 public:
  using value_type = T;
  using error_type = E;

  constexpr expected() : v_{std::in_place_index<0>} {}
  template <class U = T>
    requires(std::is_constructible_v<T, U> && !std::is_same_v<std::remove_cvref_t<U>, expected> &&
             !std::is_same_v<std::remove_cvref_t<U>, std::in_place_t> &&
             !std::is_same_v<std::remove_cvref_t<U>, unexpect_t>)
  constexpr expected(U&& v) : v_{std::in_place_index<0>, std::forward<U>(v)} {}
  template <class G>
    requires std::is_constructible_v<E, const G&>
  constexpr expected(const unexpected<G>& u) : v_{std::in_place_index<1>, u.error()} {}
  template <class... A>
  constexpr explicit expected(std::in_place_t, A&&... a) : v_{std::in_place_index<0>, std::forward<A>(a)...} {}
  template <class... A>
  constexpr explicit expected(unexpect_t, A&&... a) : v_{std::in_place_index<1>, std::forward<A>(a)...} {}

  [[nodiscard]] constexpr auto has_value() const noexcept -> bool { return v_.index() == 0; }
  constexpr explicit operator bool() const noexcept { return has_value(); }
  constexpr auto value() & -> T& { assert(has_value()); return *std::get_if<0>(&v_); }
  constexpr auto value() const& -> const T& { assert(has_value()); return *std::get_if<0>(&v_); }
  constexpr auto operator*() & -> T& { return value(); }
  constexpr auto operator*() const& -> const T& { return value(); }
  constexpr auto operator->() -> T* { return &value(); }
  constexpr auto operator->() const -> const T* { return &value(); }
  constexpr auto error() & -> E& { assert(!has_value()); return *std::get_if<1>(&v_); }
  constexpr auto error() const& -> const E& { assert(!has_value()); return *std::get_if<1>(&v_); }
};

}  // namespace starflate::compat
#endif
