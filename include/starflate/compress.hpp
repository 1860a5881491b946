// starflate::compress -- the sibling of decompress(): raw RFC 1951 (optionally zlib / gzip wrapped) out of
// the MI355X kernels.
// Thin C++23 wrapper over the C-ABI (include/starflate_hip.h); link with libstarflate_hip.so.
// Same conventions as the reference's one public function
// (/root/reference/src/decompress.hpp:63-71): non-owning spans, caller-owned buffers, no
// exceptions, a uint8_t status enum; the size comes back through expected<>.
#pragma once
#include "starflate/compat/expected.hpp"
#include "starflate/container.hpp"
#include "starflate_hip.h"

#include <cstddef>
#include <cstdint>
#include <span>
#include <utility>
#include <vector>

namespace starflate {

enum class CompressStatus : std::uint8_t {
  Success,
  InvalidArgument,
  DstTooSmall,  // dst.size() < compress_bound(src.size())
  NoDevice,     // no MI355X visible: there is no CPU fallback
  DeviceError,
  OutOfMemory,
  CommError,    // RCCL not loadable, or an RCCL call failed (gather_streams)
  Unsupported,  // the effort rests on LDS behaviour this device does not show (sfh_lds_order_check)
};

enum class BlockStrategy : std::uint8_t { Auto, Stored, Fixed, Dynamic };

/// How hard the match finder looks (== enum sfh_effort of the C-ABI, value for value).
///   Default : every other position searched, two history levels per hash bucket + the step-local candidate
///   Fast    : the newer history level only, about 3 % more output
///   Fastest : that, and no step-local candidate
///   Thorough: every position searched
///   Max     : Thorough with a second hash table keyed by seven bytes
///   Best / Ultra / Extreme: exact hash chains of depth 8 / 16 / 32 (zlib's structure) instead of the step tables
///   Recent / RecentAll: the step tables with EXACT RECENCY -- a bucket holds the latest position with the hash and the one
///             before the latest inserting step, a position's nearest earlier occurrence is its third candidate; every
///             other position searched (Recent: DEPRECATED -- Thorough's ratio at Thorough's speed on every workload
///             measured, kept for compatibility) or every position (RecentAll: the effort for real source text / machine code)
enum class Effort : std::uint8_t {
  Default = SFH_EFFORT_DEFAULT,
  Fast = SFH_EFFORT_FAST,
  Fastest = SFH_EFFORT_FASTEST,
  Thorough = SFH_EFFORT_THOROUGH,
  Max = SFH_EFFORT_MAX,
  Best = SFH_EFFORT_BEST,    // exact hash chains, the 8 most recent positions with the hash
  Ultra = SFH_EFFORT_ULTRA,  // ... the 16 most recent
  Extreme = SFH_EFFORT_EXTREME,  // ... the 32 most recent: zlib -6's own ratio
  Recent [[deprecated("no point of its own on the speed / ratio curve: use Thorough or RecentAll")]] = SFH_EFFORT_RECENT,
  RecentAll = SFH_EFFORT_RECENT_ALL,
};

struct compress_options {
  BlockStrategy strategy{BlockStrategy::Auto};
  bool final_stream{true};  // false: byte-aligned, non-final stream (a shard that is not the last)
  std::uint8_t lazy{3};     // 0..3 positions of look-ahead of the lazy match rule (sfh_options.lazy)
  bool stored_fast_path{true};  // skip the search of a chunk whose first 8 KiB are (almost) all literals
  Container container{Container::Raw};  // Zlib / Gzip: wrapper + GPU-computed Adler-32 / CRC-32 (needs final_stream)
  int device{0};
  Effort effort{Effort::Default};  // sfh_options.effort
  std::uint8_t chain_depth{0};     // sfh_options.chain_depth: with Best / Ultra / Extreme, candidates per position (0: 8 / 16 / 32)
  std::uint32_t block_bytes{0};  // bytes coded independently of what precedes them: a multiple of 32768, 0 = default
                                 // (sfh_options.block_bytes); larger compresses better, 32768 = independent DEFLATE blocks
};

inline auto compress_bound(std::size_t n, std::uint32_t block_bytes = 0) -> std::size_t { return sfh_compress_bound(n, block_bytes); }

/// What makes a stream of this library decodable in parallel (on the GPU): the first stream byte of every
/// 32 KiB segment plus the end of the last one, the strip size (no match reaches before its strip), and
/// optionally, per segment, 32 x {bit offset of the first token code of every 1024 bytes, tokens before it}.
/// Side information: the stream itself is plain DEFLATE.
struct stream_index {
  std::vector<std::uint64_t> offsets;  // segments + 1
  std::vector<std::uint32_t> regions;  // segments * 64, or empty
  std::uint32_t block_bytes{SFH_SEGMENT_BYTES};  // strip size the stream was written with
  [[nodiscard]] auto segments() const -> std::size_t { return offsets.empty() ? 0 : offsets.size() - 1; }
};

namespace detail {
inline auto to_status(int rc) -> CompressStatus {
  switch (rc) {
    case SFH_OK: return CompressStatus::Success;
    case SFH_E_DST_TOO_SMALL: return CompressStatus::DstTooSmall;
    case SFH_E_NO_DEVICE: return CompressStatus::NoDevice;
    case SFH_E_HIP: return CompressStatus::DeviceError;
    case SFH_E_NOMEM: return CompressStatus::OutOfMemory;
    case SFH_E_COMM: return CompressStatus::CommError;
    case SFH_E_UNSUPPORTED: return CompressStatus::Unsupported;
    default: return CompressStatus::InvalidArgument;
  }
}
inline auto to_c(const compress_options& o) -> sfh_options {
  sfh_options c;
  sfh_default_options(&c);
  c.strategy = static_cast<std::uint32_t>(o.strategy);
  c.final_stream = o.final_stream ? 1U : 0U;
  c.lazy = o.lazy;
  c.no_stored_fast_path = o.stored_fast_path ? 0U : 1U;
  c.container = static_cast<std::uint32_t>(o.container);
  c.block_bytes = o.block_bytes;
  c.effort = static_cast<std::uint32_t>(o.effort);
  c.chain_depth = o.chain_depth;
  return c;
}
}  // namespace detail

/// One GPU context (device scratch, stream).  Not thread-safe; distinct objects are independent.
class compressor {
  sfh_ctx* ctx_{nullptr};
  CompressStatus init_{CompressStatus::Success};

 public:
  explicit compressor(int device = 0) { init_ = detail::to_status(sfh_create(&ctx_, device)); }
  compressor(const compressor&) = delete;
  auto operator=(const compressor&) -> compressor& = delete;
  compressor(compressor&& o) noexcept : ctx_{std::exchange(o.ctx_, nullptr)}, init_{o.init_} {}
  auto operator=(compressor&& o) noexcept -> compressor& {
    if (this != &o) {
      sfh_destroy(ctx_);
      ctx_ = std::exchange(o.ctx_, nullptr);
      init_ = o.init_;
    }
    return *this;
  }
  ~compressor() { sfh_destroy(ctx_); }
  [[nodiscard]] auto status() const -> CompressStatus { return init_; }
  [[nodiscard]] auto native() const -> sfh_ctx* { return ctx_; }  // for the C-ABI entry points taking several contexts

  /// host spans: H2D, compress, D2H
  auto compress(std::span<const std::byte> src, std::span<std::byte> dst, const compress_options& opt = {})
      -> compat::expected<std::size_t, CompressStatus> {
    if (!ctx_) return compat::unexpected{init_};
    const auto c = detail::to_c(opt);
    std::size_t n = 0;
    const int rc = sfh_compress(ctx_, src.data(), src.size(), dst.data(), dst.size(), &n, &c);
    if (rc != SFH_OK) return compat::unexpected{detail::to_status(rc)};
    return n;
  }
  /// index of the last compress call on this object (with_regions: also the per-region sub-index)
  auto index(bool with_regions = true) -> compat::expected<stream_index, CompressStatus> {
    if (!ctx_) return compat::unexpected{init_};
    stream_index ix;
    const std::size_t n = sfh_index_entries(ctx_);
    if (n == 0) return compat::unexpected{CompressStatus::InvalidArgument};
    ix.offsets.resize(n);
    ix.block_bytes = sfh_last_block_bytes(ctx_);
    int rc = sfh_copy_index(ctx_, ix.offsets.data(), n, 0, nullptr);
    if (rc == SFH_OK && with_regions) {
      ix.regions.resize((n - 1) * SFH_SUBINDEX_WORDS);
      rc = sfh_copy_subindex(ctx_, ix.regions.data(), ix.regions.size(), 0, nullptr);
    }
    if (rc != SFH_OK) return compat::unexpected{detail::to_status(rc)};
    return ix;
  }
  /// decompress() on the GPU for an indexed stream: same statuses as the serial one (src/decompress.hpp:13-23),
  /// reported for the first failing segment in stream order.  dst.size() is the exact output size.
  /// A problem on the device side (no device, allocation) is DecompressStatus::Error.
  auto decompress(std::span<const std::byte> src, std::span<std::byte> dst, const stream_index& ix) -> DecompressStatus {
    if (!ctx_ || ix.offsets.size() < 2 || (!ix.regions.empty() && ix.regions.size() != ix.segments() * SFH_SUBINDEX_WORDS))
      return DecompressStatus::Error;
    std::uint32_t st = 0;
    const int rc = sfh_decompress(ctx_, src.data(), src.size(), ix.offsets.data(), ix.regions.empty() ? nullptr : ix.regions.data(),
                                  ix.segments(), dst.data(), dst.size(), ix.block_bytes, &st);
    if (rc != SFH_OK || st > 7) return DecompressStatus::Error;
    return static_cast<DecompressStatus>(st);
  }
  /// device pointers (src 16-byte aligned), optional hipStream_t
  auto compress_device(const void* d_src, std::size_t n, void* d_dst, std::size_t cap, const compress_options& opt = {},
                       void* stream = nullptr) -> compat::expected<std::size_t, CompressStatus> {
    if (!ctx_) return compat::unexpected{init_};
    const auto c = detail::to_c(opt);
    std::size_t out = 0;
    const int rc = sfh_compress_device(ctx_, d_src, n, d_dst, cap, &out, &c, stream);
    if (rc != SFH_OK) return compat::unexpected{detail::to_status(rc)};
    return out;
  }
  /// enqueue only: the stream size is left in *d_out_n, a device word (what gather_streams reads)
  auto compress_device_async(const void* d_src, std::size_t n, void* d_dst, std::size_t cap, std::uint64_t* d_out_n,
                             const compress_options& opt = {}, void* stream = nullptr) -> CompressStatus {
    if (!ctx_) return init_;
    const auto c = detail::to_c(opt);
    return detail::to_status(sfh_compress_device_async(ctx_, d_src, n, d_dst, cap, d_out_n, &c, stream));
  }

  /// One process per GPU: what every rank of an RCCL communicator calls after compressing its shard (whole strips,
  /// final_stream = false on every rank but the last).  The byte-aligned streams are put back to back, in rank order, at
  /// d_out + base on `root`: one ncclAllGather of the sizes, then one point-to-point transfer per peer (no ring, no
  /// all-reduce) -- legal because a match only has to stay inside the bytes already written
  /// (/root/reference/src/decompress.cpp:178) and blocks simply follow one another until BFINAL (:410-415).
  /// nccl_comm: an ncclComm_t.  base / cap: the root's (bytes already in d_out, its capacity), the same on every rank.
  /// Returns the end of the concatenation; `sizes` (optional) receives every rank's stream size.
  auto gather_streams(void* nccl_comm, int root, const void* d_stream, const std::uint64_t* d_size, void* d_out, std::uint64_t base,
                      std::uint64_t cap, void* stream = nullptr, std::vector<std::uint64_t>* sizes = nullptr)
      -> compat::expected<std::uint64_t, CompressStatus> {
    if (!ctx_) return compat::unexpected{init_};
    int nranks = 0, rank = 0;
    if (const int rc = sfh_comm_ranks(nccl_comm, &nranks, &rank); rc != SFH_OK) return compat::unexpected{detail::to_status(rc)};
    std::vector<std::uint64_t> local(static_cast<std::size_t>(nranks));
    std::uint64_t end = 0;
    const int rc = sfh_gather_streams(ctx_, nccl_comm, root, d_stream, d_size, d_out, base, cap, local.data(), &end, stream);
    if (rc != SFH_OK) return compat::unexpected{detail::to_status(rc)};
    if (sizes != nullptr) *sizes = std::move(local);
    return end;
  }
};

/// One process, several GPUs: contiguous shards of `src` on the given compressors (one per device), one stream
/// in `dst`, bit-identical to a single compress() call.
inline auto compress(std::span<compressor* const> gpus, std::span<const std::byte> src, std::span<std::byte> dst,
                     const compress_options& opt = {}) -> compat::expected<std::size_t, CompressStatus> {
  std::vector<sfh_ctx*> h;
  for (const auto* g : gpus) {
    if (g == nullptr || g->native() == nullptr) return compat::unexpected{g != nullptr ? g->status() : CompressStatus::InvalidArgument};
    h.push_back(g->native());
  }
  const auto c = detail::to_c(opt);
  std::size_t n = 0;
  const int rc = sfh_compress_multi(h.data(), static_cast<int>(h.size()), src.data(), src.size(), dst.data(), dst.size(), &n, &c);
  if (rc != SFH_OK) return compat::unexpected{detail::to_status(rc)};
  return n;
}

namespace detail {
inline auto thread_compressors() -> std::vector<std::pair<int, compressor>>& {
  thread_local std::vector<std::pair<int, compressor>> cache;
  return cache;
}
/// The calling thread's context for `device`, created on first use and kept until the thread exits or calls
/// release_thread_compressor(): the free function below does not pay for a stream, events and device scratch on
/// every call.  A context holds device scratch sized by its largest call (2.3 bytes per input byte, at most 2.3 GiB).
inline auto thread_compressor(int device) -> compressor& {
  auto& cache = thread_compressors();
  for (auto& e : cache)
    if (e.first == device) return e.second;
  cache.emplace_back(device, compressor{device});
  return cache.back().second;
}
}  // namespace detail

/// Frees the calling thread's cached context(s) -- stream, events, device scratch -- for `device`, or for every
/// device with -1.  The next free-function compress() on this thread makes a new one.  Returns how many were freed.
inline auto release_thread_compressor(int device = -1) -> std::size_t {
  auto& cache = detail::thread_compressors();
  const auto before = cache.size();
  std::erase_if(cache, [device](const auto& e) { return device < 0 || e.first == device; });
  return before - cache.size();
}

/// Compresses `src` into `dst` (dst.size() >= compress_bound(src.size())); returns the stream size.
/// Re-entrant like the reference's decompress() (src/decompress.hpp:63-71): each thread keeps its own context per device.
inline auto compress(std::span<const std::byte> src, std::span<std::byte> dst, const compress_options& opt = {})
    -> compat::expected<std::size_t, CompressStatus> {
  return detail::thread_compressor(opt.device).compress(src, dst, opt);
}

}  // namespace starflate
