// Times the Engine-shaped call of the C++ adapter (libsbn_amd/csrc/host/engine.hpp):
// Engine::Gradients(const UnrootedTreeCollection&, params, rescaling) ->
// std::vector<PhyloGradient>, the signature of /root/reference/src/engine.cpp:78-84 -- host tree
// collections in, host vectors of PhyloGradient (a std::map of std::vector<double> per tree)
// out.  bench.py builds it with g++ against libmi_phylo.so and runs it as a child process on
// the headline batch (leg "adapter" of bench_also.json).
//
//   engine_bench <fasta> <newick> <branch_lengths.f64> <trees> <steps> <warmup> <out.f64>
//
// branch_lengths.f64: [trees][2n-2] doubles written by bench.py (the batch of the headline
// line, so that the results can be compared bit for bit); the topologies of <newick> are
// cycled.  Prints one JSON object; out.f64 receives [trees][1 + 2n-1 + 1] doubles (logL,
// branch gradient, site gradient) of the last step.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "../libsbn_amd/csrc/host/engine.hpp"

using namespace mihost;

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: engine_bench fasta newick bl.f64 trees steps warmup out.f64\n");
    return 2;
  }
  try {
    const size_t T = std::strtoul(argv[4], nullptr, 10);
    const int steps = std::atoi(argv[5]), warmup = std::atoi(argv[6]);
    auto parsed = TreeCollection::ParseNewickFile(argv[2]);
    SitePattern pattern(Alignment::ReadFasta(argv[1]), parsed.taxon_names_);
    const size_t n = pattern.SequenceCount(), N = 2 * n - 1, B = 2 * n - 2;
    std::vector<double> bl(T * B);
    {
      std::ifstream in(argv[3], std::ios::binary);
      in.read(reinterpret_cast<char*>(bl.data()), static_cast<std::streamsize>(bl.size() * 8));
      if (!in) Failwith("short branch-length file");
    }
    UnrootedTreeCollection trees(T);
    for (size_t t = 0; t < T; t++) {
      trees[t] = parsed.trees_[t % parsed.TreeCount()];
      trees[t].branch_lengths.assign(bl.begin() + t * B, bl.begin() + (t + 1) * B);
    }
    Engine engine(EngineSpecification{1, {}, true}, {"JC69", "weibull+4", "strict"}, pattern);
    ParamMatrix params(T, engine.ParameterCount());
    params.SetBlock(0, engine.ParameterCount(), {1.0});
    std::vector<PhyloGradient> last;
    for (int i = 0; i < warmup; i++) last = engine.Gradients(trees, params, false);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) last = engine.Gradients(trees, params, false);
    const double sec =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> out;
    out.reserve(T * (N + 2));
    for (const auto& g : last) {
      out.push_back(g.log_likelihood_);
      const auto& b = g.gradient_.at("branch_lengths");
      out.insert(out.end(), b.begin(), b.end());
      out.push_back(g.gradient_.at("site_model")[0]);
    }
    std::ofstream(argv[7], std::ios::binary)
        .write(reinterpret_cast<const char*>(out.data()),
               static_cast<std::streamsize>(out.size() * 8));
    std::printf("{\"trees\": %zu, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, "
                "\"trees_per_s\": %.1f, \"logL0\": %.17g}\n",
                T, steps, warmup, 1e3 * sec / steps, T * steps / sec, last[0].log_likelihood_);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "engine_bench: %s\n", e.what());
    return 1;
  }
  return 0;
}
