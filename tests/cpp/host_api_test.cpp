// Host C++23 API checks, CPU only.  The cases restate, as data, what the reference's own
// tests assert (boost.ut is not available, so a 10-line checker stands in):
//   huffman/test/{bit,code,bit_span,table_from_*,table_find_code,decode}_test.cpp
//   src/test/decompress_test.cpp
// argv[1] = tests/golden directory; argv[2] (optional) = directory holding starfleet.html.zlib / .gz /
// .named.gz made by zlib / gzip at test time (tests/test_cpp_host_api.py).
#include "starflate/container.hpp"
#include "starflate/decompress.hpp"
#include "starflate/huffman/huffman.hpp"
#include "starflate/huffman/lookup_decoder.hpp"

#include <cstdio>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

namespace huffman = starflate::huffman;
using namespace huffman::literals;
using starflate::DecompressStatus;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                   \
  do {                                                                \
    ++g_checks;                                                       \
    if (!(cond)) {                                                    \
      ++g_fail;                                                       \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
    }                                                                 \
  } while (0)

static auto read_file(const std::string& path) -> std::vector<std::byte> {
  std::ifstream f{path, std::ios::binary};
  std::vector<char> c((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<std::byte> b(c.size());
  for (std::size_t i = 0; i < c.size(); ++i) b[i] = static_cast<std::byte>(c[i]);
  return b;
}

template <class T>
static auto str(const T& v) -> std::string {
  std::ostringstream ss;
  ss << v;
  return ss.str();
}

static void test_bit_and_code() {
  CHECK(bool(huffman::bit{1}) && !bool(huffman::bit{0}) && bool(huffman::bit{'1'}) && bool(1_b) && !bool(0_b));
  CHECK(str(huffman::bit{true}) == "1");
  constexpr auto c = 110_c;
  static_assert(c.bitsize() == 3 && c.value() == 6);
  CHECK(str(c) == "110" && str(00101_c) == "00101");
  auto d = huffman::code{};
  d << 1_b << 0_b << 1_b;  // right pad
  CHECK(d == 101_c);
  1_b >> d;                // left pad
  CHECK(d == 1101_c);
  CHECK((1_c < 00_c) && (01_c < 10_c));  // ordered by (bitsize, value)
  CHECK(sizeof(huffman::code) == 16);
}

static void test_bit_span() {
  constexpr auto data = huffman::byte_array(0b10101010, 0xff);
  std::string s;
  for (auto b : huffman::bit_span{data}) s += static_cast<char>(b);
  CHECK(s == "0101010111111111");
  huffman::bit_span sp{data};
  sp.consume(3);
  CHECK(std::ranges::size(sp) == 13 && bool(*sp.begin()) == true);
  sp.consume_to_byte_boundary();
  CHECK(std::ranges::size(sp) == 8 && sp.byte_data() == data.data() + 1);
  constexpr auto le = huffman::byte_array(0xAA, 0x55, 0x01);
  huffman::bit_span p{le};
  CHECK(p.pop_16() == 0x55AA && p.pop_8() == 0x01 && p.empty());
  huffman::bit_span off{data.data(), 5, 2};
  std::string t;
  for (auto b : off) t += static_cast<char>(b);
  CHECK(t == "01010");
}

static void test_table_from_frequencies() {
  const auto freq = std::vector<std::pair<char, std::size_t>>{{'e', 100}, {'n', 20}, {'x', 1}, {'i', 40}, {'q', 3}};
  const auto with_eot = huffman::table{freq, char{4}};
  CHECK(str(with_eot) ==
        "Bits\tCode\tValue\tSymbol\n1\t0\t0\t`e`\n2\t10\t2\t`i`\n3\t110\t6\t`n`\n4\t1110\t14\t`q`\n"
        "5\t11110\t30\t`\4`\n5\t11111\t31\t`x`\n");
  const auto no_eot = huffman::table{freq};
  CHECK(str(no_eot) ==
        "Bits\tCode\tValue\tSymbol\n1\t0\t0\t`e`\n2\t10\t2\t`i`\n3\t110\t6\t`n`\n4\t1110\t14\t`q`\n4\t1111\t15\t`x`\n");
  // five equal weights: a,b three bits; c,d,e two bits (SURVEY.md 3.3 probe)
  const auto eq = huffman::table{std::vector<std::pair<char, std::size_t>>{{'a', 1}, {'b', 1}, {'c', 1}, {'d', 1}, {'e', 1}}};
  std::string lens;
  for (const auto& e : eq) lens += std::string(1, e.symbol) + std::to_string(e.bitsize());
  CHECK(lens == "c2d2e2a3b3");
  // 24 Fibonacci weights: no length limit in this constructor
  std::vector<std::pair<int, std::size_t>> fib;
  std::size_t a = 1, b = 1;
  for (int i = 0; i < 24; ++i) { fib.emplace_back(i, a); const auto n = a + b; a = b; b = n; }
  std::uint8_t longest = 0;
  for (const auto& e : huffman::table{fib}) longest = std::max(longest, e.bitsize());
  CHECK(longest == 23);
}

static void test_table_from_data_and_edges() {
  const auto t = huffman::table{std::string_view{"eeeeeeeeiiiinnq"}, char{4}};
  CHECK(t.begin()->symbol == 'e' && t.begin()->bitsize() == 1 && t.size() == 5);
  const auto one = huffman::table{std::string_view{"aaaa"}};
  CHECK(one.size() == 1 && one.begin()->bitsize() == 1 && one.begin()->value() == 0);
  const huffman::table<char> empty{};
  CHECK(empty.begin() == empty.end());
}

static void test_table_from_symbol_bitsize() {
  // RFC 1951 3.2.2 example 1
  static constexpr auto t1 = huffman::table<char, 4>{huffman::symbol_bitsize, {{'A', 2}, {'B', 1}, {{'C', 'D'}, 3}}};
  static constexpr auto e1 = huffman::table{huffman::table_contents,
                                            {std::pair{0_c, 'B'}, {10_c, 'A'}, {110_c, 'C'}, {111_c, 'D'}}};
  static_assert(std::ranges::equal(t1, e1));
  // example 2
  static constexpr auto t2 = huffman::table<char, 8>{huffman::symbol_bitsize, {{{'A', 'E'}, 3}, {'F', 2}, {{'G', 'H'}, 4}}};
  static constexpr auto e2 = huffman::table{huffman::table_contents,
      {std::pair{00_c, 'F'}, {010_c, 'A'}, {011_c, 'B'}, {100_c, 'C'}, {101_c, 'D'}, {110_c, 'E'}, {1110_c, 'G'}, {1111_c, 'H'}}};
  static_assert(std::ranges::equal(t2, e2));
  CHECK(std::ranges::equal(t1, e1) && std::ranges::equal(t2, e2));
  // RFC 1951 3.2.6 fixed literal/length code
  using span = huffman::symbol_span<std::uint16_t>;
  const auto fixed = huffman::table<std::uint16_t, 288>{
      huffman::symbol_bitsize, {{span{0, 143}, 8}, {span{144, 255}, 9}, {span{256, 279}, 7}, {span{280, 287}, 8}}};
  auto code_of = [&](std::uint16_t s) {
    for (const auto& e : fixed)
      if (e.symbol == s) return static_cast<const huffman::code&>(e);
    return huffman::code{};
  };
  CHECK(code_of(0) == huffman::code(8, 0x30) && code_of(143) == huffman::code(8, 0xBF));
  CHECK(code_of(144) == huffman::code(9, 0x190) && code_of(255) == huffman::code(9, 0x1FF));
  CHECK(code_of(256) == huffman::code(7, 0x00) && code_of(279) == huffman::code(7, 0x17));
  CHECK(code_of(280) == huffman::code(8, 0xC0) && code_of(287) == huffman::code(8, 0xC7));
}

static constexpr auto kTable = huffman::table{
    huffman::table_contents,
    {std::pair{0_c, 'e'}, {10_c, 'i'}, {110_c, 'n'}, {1110_c, 'q'}, {11110_c, '\4'}, {11111_c, 'x'}}};

static void test_find() {
  // north_star's name for the type (the reference keeps it as a parameter name, huffman/src/decode.hpp:25,85)
  static_assert(std::is_same_v<huffman::code_table<char, 5>, huffman::table<char, 5>>);
  static_assert(std::is_same_v<huffman::code_table<std::uint16_t>, huffman::table<std::uint16_t, std::dynamic_extent>>);
  static_assert(kTable.find(0_c).value()->symbol == 'e');
  static_assert(kTable.find(11111_c).value()->symbol == 'x');
  CHECK(kTable.find(10_c).value()->symbol == 'i' && kTable.find(1110_c).value()->symbol == 'q');
  // a prefix of longer codes: error = first entry with a longer bitsize
  const auto r1 = kTable.find(1_c);
  CHECK(!r1 && r1.error()->symbol == 'i' && r1.error()->bitsize() == 2);
  const auto r2 = kTable.find(11_c);
  CHECK(!r2 && r2.error()->symbol == 'n');
  const auto r3 = kTable.find(111_c, r2.error());
  CHECK(!r3 && r3.error()->symbol == 'q');
  // beyond the longest code: end()
  CHECK(kTable.find(111111_c).error() == kTable.end());
  CHECK(kTable.find(11110_c, kTable.find(1111_c).error()).value()->symbol == '\4');
}

template <std::size_t N>
static auto dec(const std::array<std::byte, N>& bytes, std::size_t drop_bits = 0) -> std::string {
  std::string out;
  huffman::decode(kTable, huffman::bit_span{bytes.data(), N * CHAR_BIT - drop_bits}, std::back_inserter(out));
  return out;
}

static void test_decode() {
  std::string none;
  huffman::decode(kTable, huffman::bit_span{}, std::back_inserter(none));
  CHECK(none.empty());
  CHECK(dec(huffman::byte_array(0b11111011)) == "nx");
  CHECK(dec(huffman::byte_array(0b11111011, 0b00010111)) == "nxqiee");
  CHECK(dec(huffman::byte_array(0b11111011, 0b00010111), 2) == "nxqi");
  CHECK(dec(huffman::byte_array(0b10111110, 0b11000001, 0b01011111)) == "exeneeeexni");
  CHECK(dec(huffman::byte_array(0b10111110, 0b11000001, 0b01011111, 0b00110111, 0b01101001, 0b00111101), 1) ==
        "exeneeeexniqneieini\4");
  constexpr auto one_e = huffman::byte_array(0);
  const auto r = huffman::decode_one(kTable, huffman::bit_span{one_e.data(), 1, 7});
  CHECK(r.has_value() && r.symbol() == 'e' && r.encoded_size() == 1);
}

static void test_decompress(const std::string& golden) {
  using starflate::decompress;
  namespace detail = starflate::detail;
  {
    huffman::bit_span empty{nullptr, 0, 0};
    CHECK(detail::read_header(empty).error() == DecompressStatus::InvalidBlockHeader);
    constexpr auto bad = huffman::byte_array(0b111);
    huffman::bit_span b{bad};
    CHECK(detail::read_header(b).error() == DecompressStatus::InvalidBlockHeader);
    constexpr auto fixed = huffman::byte_array(0b010);
    huffman::bit_span f{fixed};
    const auto h = detail::read_header(f);
    CHECK(h && !h->final && h->type == detail::BlockType::FixedHuffman);
    constexpr auto stored = huffman::byte_array(0b001);
    huffman::bit_span s{stored};
    const auto g = detail::read_header(s);
    CHECK(g && g->final && g->type == detail::BlockType::NoCompression);
  }
  CHECK(decompress(std::span<const std::byte>{}, std::span<std::byte>{}) == DecompressStatus::InvalidBlockHeader);
  {
    constexpr auto comp = huffman::byte_array(0b000, 4, 0, ~4, ~0, 'r', 'o', 's', 'e', 0b001, 3, 0, ~3, ~0, 'b', 'u', 'd');
    constexpr auto want = huffman::byte_array('r', 'o', 's', 'e', 'b', 'u', 'd');
    std::array<std::byte, want.size()> dst{};
    const std::span<const std::byte> src{comp};
    CHECK(decompress(src, std::span<std::byte>{dst.data(), dst.size() - 1}) == DecompressStatus::DstTooSmall);
    CHECK(decompress(src.subspan(0, 5), std::span<std::byte>{dst}) == DecompressStatus::SrcTooSmall);
    CHECK(decompress(src, std::span<std::byte>{dst}) == DecompressStatus::Success && std::ranges::equal(dst, want));
    CHECK(decompress(comp, std::span<std::byte>{dst}) == DecompressStatus::Success);  // range overload
  }
  {
    const auto want = read_file(golden + "/starfleet.html");
    for (const char* name : {"/starfleet.html.fixed", "/starfleet.html.dynamic"}) {
      const auto comp = read_file(golden + name);
      std::vector<std::byte> dst(want.size());
      CHECK(!comp.empty() && decompress(comp, dst) == DecompressStatus::Success && dst == want);
    }
  }
  {
    auto buf = huffman::byte_array(1, 2, 0, 0, 0, 0);
    const auto d = std::span<std::byte>{buf}.subspan(2);
    detail::copy_from_before(2, d.begin(), 3);
    CHECK(buf == huffman::byte_array(1, 2, 1, 2, 1, 0));
  }
  static_assert(static_cast<int>(DecompressStatus::InvalidDistance) == 7 && static_cast<int>(DecompressStatus::DstTooSmall) == 4);
}

// lookup_decoder must give exactly what the per-bit decode_one gives: every 16-bit pattern, at every length the
// input may be cut to, for complete, incomplete and long-code tables
template <std::uint8_t Bits, class Table>
static void check_lookup(const Table& t) {
  const huffman::lookup_decoder<std::uint16_t, Bits> fast{t};
  for (std::uint32_t pat = 0; pat < (1U << 16U); pat += 7) {
    const auto bytes = huffman::byte_array(pat & 0xFFU, (pat >> 8U) & 0xFFU, 0xA5);
    for (const std::size_t nbits : {std::size_t{0}, std::size_t{1}, std::size_t{5}, std::size_t{9}, std::size_t{16}, std::size_t{21}}) {
      for (const std::uint8_t off : {std::uint8_t{0}, std::uint8_t{3}}) {
        const huffman::bit_span bits{bytes.data(), nbits, off};
        const auto a = huffman::decode_one(t, bits);
        const auto b = fast.decode_one(t, bits);
        if (a.has_value() != b.has_value() || (a.has_value() && (a.symbol() != b.symbol() || a.encoded_size() != b.encoded_size()))) {
          CHECK(false);
          return;
        }
      }
    }
  }
  CHECK(true);
}

static void test_lookup_decoder() {
  using pairs = std::vector<std::pair<huffman::symbol_span<std::uint16_t>, std::uint8_t>>;
  using sspan = huffman::symbol_span<std::uint16_t>;
  // RFC 1951 3.2.2 example, the fixed literal/length code, a code with 12..15-bit members, an incomplete code
  const pairs rfc{{sspan{0, 4}, 3}, {sspan{5}, 2}, {sspan{6}, 4}, {sspan{7}, 4}};
  const pairs fixed{{sspan{0, 143}, 8}, {sspan{144, 255}, 9}, {sspan{256, 279}, 7}, {sspan{280, 287}, 8}};
  const pairs deep{{sspan{0}, 1}, {sspan{1}, 2}, {sspan{2}, 3}, {sspan{3}, 4}, {sspan{4}, 5}, {sspan{5}, 6}, {sspan{6}, 7}, {sspan{7}, 8},
                   {sspan{8}, 9}, {sspan{9}, 10}, {sspan{10}, 11}, {sspan{11}, 12}, {sspan{12}, 13}, {sspan{13}, 14}, {sspan{14, 15}, 15}};
  const pairs incomplete{{sspan{3}, 2}, {sspan{9}, 5}};
  for (const auto* p : {&rfc, &fixed, &deep, &incomplete}) {
    const huffman::table<std::uint16_t> t{huffman::symbol_bitsize, *p};
    check_lookup<9>(t);
    check_lookup<7>(t);
    check_lookup<1>(t);
  }
  const std::array<std::byte, 2> two{std::byte{0xFF}, std::byte{0x01}};
  CHECK((huffman::bit_span{two.data(), 16}.peek(9) == 0x1FFU && huffman::bit_span{two.data(), 5, 6}.peek(9) == 0x07U));
}

static auto bytes_of(const char* s) -> std::vector<std::byte> {
  std::vector<std::byte> b;
  for (; *s != 0; ++s) b.push_back(static_cast<std::byte>(*s));
  return b;
}

static void test_container(const std::string& golden, const std::string& wrapped) {
  using starflate::Container;
  using starflate::decompress;
  // published check values: CRC-32("123456789") and Adler-32("Wikipedia")
  CHECK(starflate::crc32(bytes_of("123456789")) == 0xCBF43926U);
  CHECK(starflate::adler32(bytes_of("Wikipedia")) == 0x11E60398U);
  CHECK(starflate::crc32({}) == 0U && starflate::adler32({}) == 1U);
  {
    const std::vector<std::byte> ff(70000, std::byte{0xFF});  // no 32-bit overflow inside a 5552-byte run
    std::uint64_t a = 1, b = 0;
    for (std::size_t i = 0; i < ff.size(); ++i) { a = (a + 255) % 65521; b = (b + a) % 65521; }
    CHECK(starflate::adler32(ff) == ((b << 16U) | a));
  }
  if (wrapped.empty()) return;
  const auto want = read_file(golden + "/starfleet.html");
  for (const auto& [name, kind] : {std::pair{"/starfleet.html.zlib", Container::Zlib}, std::pair{"/starfleet.html.gz", Container::Gzip},
                                   std::pair{"/starfleet.html.named.gz", Container::Gzip}}) {
    auto comp = read_file(wrapped + name);
    std::vector<std::byte> dst(want.size());
    CHECK(!comp.empty() && decompress(comp, dst, kind) == DecompressStatus::Success && dst == want);
    comp[comp.size() - (kind == Container::Zlib ? 1U : 5U)] ^= std::byte{1};  // checksum byte
    CHECK(decompress(comp, dst, kind) == DecompressStatus::Error);
    comp[comp.size() - (kind == Container::Zlib ? 1U : 5U)] ^= std::byte{1};
    comp[0] ^= std::byte{0x10};  // header
    CHECK(decompress(comp, dst, kind) == DecompressStatus::Error);
    comp[0] ^= std::byte{0x10};
    CHECK(decompress(std::span<const std::byte>{comp}.first(5), dst, kind) == DecompressStatus::SrcTooSmall);
    if (kind == Container::Gzip) {
      std::vector<std::byte> small(want.size() - 1);
      CHECK(decompress(comp, small, kind) == DecompressStatus::DstTooSmall);
    }
  }
  {
    const auto raw = read_file(golden + "/starfleet.html.dynamic");
    std::vector<std::byte> dst(want.size());
    CHECK(decompress(raw, dst, Container::Raw) == DecompressStatus::Success && dst == want);
  }
}

// LSB-first bit writer for hand-made streams
struct bit_writer {
  std::vector<std::byte> out;
  std::size_t nbits = 0;
  void put(unsigned v, unsigned n) {
    for (unsigned i = 0; i < n; ++i, ++nbits) {
      if (nbits % 8 == 0) out.push_back(std::byte{0});
      if ((v >> i) & 1U) out.back() |= static_cast<std::byte>(1U << (nbits % 8));
    }
  }
};

// Untrusted input must come back as a status, never as an abort or a table of wrapped codes: over-subscribed
// code lengths (the reference asserts on them, huffman/src/code.hpp), then a seeded bit-flip fuzz of valid streams
// (this test is built without NDEBUG, and under ASan + UBSan in the second configuration).
static void test_untrusted_input(const std::string& golden) {
  {
    // dynamic block whose code-length code has three codes of length 1 (Kraft sum 1.5)
    bit_writer w;
    w.put(1, 1); w.put(2, 2);            // BFINAL, BTYPE = 10
    w.put(0, 5); w.put(0, 5); w.put(0, 4);  // HLIT = 257, HDIST = 1, HCLEN = 4 -> lengths for symbols 16, 17, 18, 0
    w.put(1, 3); w.put(1, 3); w.put(1, 3); w.put(0, 3);
    w.put(0xFFFF, 16); w.put(0xFFFF, 16);
    std::vector<std::byte> dst(64);
    CHECK(starflate::decompress(w.out, dst) == DecompressStatus::Error);
  }
  {
    // a complete code-length code (symbols 1 and 2, one bit each) spelling over-subscribed literal/length lengths:
    // 257 lengths of 1 (HLIT = 0)
    bit_writer w;
    w.put(1, 1); w.put(2, 2);
    w.put(0, 5); w.put(0, 5); w.put(14, 4);  // HCLEN = 18: order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1
    const unsigned order[18] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1};
    for (const unsigned sym : order) w.put((sym == 1 || sym == 2) ? 1U : 0U, 3);
    for (int i = 0; i < 257 + 1; ++i) w.put(0, 1);  // code 0 = symbol 1: every length is 1
    std::vector<std::byte> dst(64);
    CHECK(starflate::decompress(w.out, dst) == DecompressStatus::Error);
  }
  for (const char* name : {"/starfleet.html.dynamic", "/starfleet.html.fixed"}) {
    const auto good = read_file(golden + name);
    const auto html = read_file(golden + "/starfleet.html");
    std::vector<std::byte> dst(html.size());
    std::uint64_t rng = 0x9E3779B97F4A7C15ULL;
    const auto next = [&] { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    int ok = 0;
    for (int it = 0; it < 3000; ++it) {
      auto bad = good;
      const int flips = 1 + static_cast<int>(next() % 3);
      // bias towards the first block header: that is where the code lengths live
      for (int f = 0; f < flips; ++f) bad[next() % (it % 2 ? 200 : bad.size())] ^= static_cast<std::byte>(1U << (next() % 8));
      const auto st = starflate::decompress(bad, dst);
      ok += st == DecompressStatus::Success;
      CHECK(static_cast<unsigned>(st) <= 7);
    }
    CHECK(ok < 3000);
  }
}

auto main(int argc, char** argv) -> int {
  const std::string golden = argc > 1 ? argv[1] : "tests/golden";
  test_container(golden, argc > 2 ? argv[2] : "");
  test_lookup_decoder();
  test_bit_and_code();
  test_bit_span();
  test_table_from_frequencies();
  test_table_from_data_and_edges();
  test_table_from_symbol_bitsize();
  test_find();
  test_decode();
  test_decompress(golden);
  test_untrusted_input(golden);
  std::printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
