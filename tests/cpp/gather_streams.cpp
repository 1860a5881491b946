// GPU test of the C++23 multi-GPU binding: compressor::compress_device_async + compressor::gather_streams over a REAL RCCL
// communicator (one rank: the box has one GPU).  The HIP runtime and RCCL are reached through dlopen, as a host program
// that links only libstarflate_hip.so would reach them.  argv[1] = tests/golden.
#include "starflate/compress.hpp"
#include "starflate/decompress.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

namespace {
template <class F>
auto sym(void* h, const char* name) -> F { return reinterpret_cast<F>(dlsym(h, name)); }
auto open_first(std::initializer_list<const char*> names) -> void* {
  for (const char* n : names)
    if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) return h;
  for (const char* n : names)
    if (void* h = dlopen(n, RTLD_NOW)) return h;
  return nullptr;
}
}  // namespace

auto main(int argc, char** argv) -> int {
  using namespace starflate;
  const std::string golden = argc > 1 ? argv[1] : "tests/golden";
  std::ifstream f{golden + "/starfleet.html", std::ios::binary};
  const std::vector<char> html((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  compressor gpu{0};
  if (gpu.status() != CompressStatus::Success) { std::printf("no device\n"); return 2; }
  void* hip = open_first({"libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"});
  const char* env = std::getenv("SFH_RCCL_LIB");
  void* rccl = open_first({env ? env : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"});
  if (!hip || !rccl) { std::printf("hip %p rccl %p: %s\n", hip, rccl, dlerror()); return 3; }
  const auto hipMalloc = sym<int (*)(void**, std::size_t)>(hip, "hipMalloc");
  const auto hipMemcpy = sym<int (*)(void*, const void*, std::size_t, int)>(hip, "hipMemcpy");
  const auto hipMemset = sym<int (*)(void*, int, std::size_t)>(hip, "hipMemset");
  const auto hipDeviceSynchronize = sym<int (*)()>(hip, "hipDeviceSynchronize");
  const auto hipFree = sym<int (*)(void*)>(hip, "hipFree");
  const auto commInitAll = sym<int (*)(void**, int, const int*)>(rccl, "ncclCommInitAll");
  const auto commDestroy = sym<int (*)(void*)>(rccl, "ncclCommDestroy");
  if (!hipMalloc || !hipMemcpy || !hipMemset || !hipDeviceSynchronize || !hipFree || !commInitAll || !commDestroy) { std::printf("symbols\n"); return 3; }
  int fail = 0;
  void* comm = nullptr;
  const int dev0 = 0;
  if (commInitAll(&comm, 1, &dev0) != 0) { std::printf("ncclCommInitAll failed\n"); return 4; }
  int nranks = 0, rank = -1;
  if (sfh_comm_ranks(comm, &nranks, &rank) != SFH_OK || nranks != 1 || rank != 0) { std::printf("sfh_comm_ranks\n"); ++fail; }

  const std::size_t n = html.size(), bound = compress_bound(n), base = 24;
  void *d_src = nullptr, *d_shard = nullptr, *d_out = nullptr, *d_size = nullptr;
  if (hipMalloc(&d_src, n + 16) || hipMalloc(&d_shard, bound) || hipMalloc(&d_out, base + bound) || hipMalloc(&d_size, 8)) { std::printf("hipMalloc\n"); return 5; }
  hipMemcpy(d_src, html.data(), n, 1 /* hipMemcpyHostToDevice */);
  hipMemset(d_out, 0, base + bound);
  compress_options opt;  // the only (= last) rank of the communicator: final stream
  if (gpu.compress_device_async(d_src, n, d_shard, bound, static_cast<std::uint64_t*>(d_size), opt) != CompressStatus::Success) { std::printf("async compress\n"); ++fail; }
  std::vector<std::uint64_t> sizes;
  const auto end = gpu.gather_streams(comm, 0, d_shard, static_cast<const std::uint64_t*>(d_size), d_out, base, base + bound, nullptr, &sizes);
  hipDeviceSynchronize();
  if (!end || sizes.size() != 1 || *end != base + sizes[0]) {
    std::printf("gather_streams: %d\n", end ? 0 : static_cast<int>(end.error()));
    ++fail;
  } else {
    std::vector<std::byte> stream(static_cast<std::size_t>(*end)), back(n);
    hipMemcpy(stream.data(), d_out, stream.size(), 2 /* hipMemcpyDeviceToHost */);
    for (std::size_t k = 0; k < base; ++k)
      if (stream[k] != std::byte{0}) { std::printf("bytes before base touched\n"); ++fail; break; }
    const auto st = decompress(std::span{stream}.subspan(base), back);
    if (st != DecompressStatus::Success || std::memcmp(back.data(), html.data(), n) != 0) { std::printf("round trip: status %d\n", static_cast<int>(st)); ++fail; }
    // the gathered bytes are the bytes of the synchronous call
    std::vector<std::byte> direct(bound);
    const auto nd = gpu.compress(std::span{reinterpret_cast<const std::byte*>(html.data()), n}, direct, opt);
    if (!nd || *nd != sizes[0] || std::memcmp(direct.data(), stream.data() + base, *nd) != 0) { std::printf("differs from compress()\n"); ++fail; }
    // too small for the gathered streams: refused, nothing posted
    const auto small = gpu.gather_streams(comm, 0, d_shard, static_cast<const std::uint64_t*>(d_size), d_out, base, base + sizes[0] - 1);
    if (small || small.error() != CompressStatus::DstTooSmall) { std::printf("expected DstTooSmall\n"); ++fail; }
  }
  hipFree(d_src); hipFree(d_shard); hipFree(d_out); hipFree(d_size);
  commDestroy(comm);
  std::printf("gather_streams: %d failures\n", fail);
  return fail ? 1 : 0;
}
