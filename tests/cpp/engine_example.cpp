// The reference's hot-path doctests (src/unrooted_sbn_instance.hpp:206-257,
// src/rooted_sbn_instance.hpp:246-286) written against the C++ Engine adapter
// (libsbn_amd/csrc/host/engine.hpp).  Exit code 0 = all checks passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>

#include "../../libsbn_amd/csrc/host/engine.hpp"

using namespace mihost;

static int failures = 0;
#define CHECK_LT(a, b)                                                              \
  do {                                                                              \
    if (!((a) < (b))) {                                                             \
      std::printf("CHECK failed %s:%d: %s = %.12g !< %.12g\n", __FILE__, __LINE__, #a, \
                  (double)(a), (double)(b));                                        \
      failures++;                                                                   \
    }                                                                               \
  } while (0)

int main(int argc, char** argv) {
  const std::string data = argc > 1 ? argv[1] : "tests/golden/data";
  const PhyloModelSpecification simple{"JC69", "constant", "strict"};
  {  // hello
    auto trees = TreeCollection::ParseNewickFile(data + "/hello.nwk");
    SitePattern pattern(Alignment::ReadFasta(data + "/hello.fasta"), trees.taxon_names_);
    Engine engine(EngineSpecification{2, {}, true}, simple, pattern);
    ParamMatrix params(trees.TreeCount(), engine.ParameterCount());
    for (double ll : engine.LogLikelihoods(trees.trees_, params, false))
      CHECK_LT(std::fabs(ll - -84.852358), 0.000001);
  }
  {  // DS1 x 10, JC69: log-likelihoods (pybeagle) and sorted gradient of the last tree
    auto trees = TreeCollection::ParseNexusFile(data + "/DS1.subsampled_10.t");
    SitePattern pattern(Alignment::ReadFasta(data + "/DS1.fasta"), trees.taxon_names_);
    const double pybeagle[10] = {-14582.995273982739, -6911.294207416366, -6916.880235529542,
                                 -6904.016888831189,  -6915.055570693576, -6915.50496696512,
                                 -6910.958836661867,  -6909.02639968063,  -6912.967861935749,
                                 -6910.7871105783515};
    for (bool tip_states : {false, true}) {
      Engine engine(EngineSpecification{2, {}, tip_states}, simple, pattern);
      ParamMatrix params(trees.TreeCount(), engine.ParameterCount());
      for (bool rescaling : {false, true}) {
        const auto ll = engine.LogLikelihoods(trees.trees_, params, rescaling);
        const auto gradients = engine.Gradients(trees.trees_, params, rescaling);
        for (size_t i = 0; i < ll.size(); i++) {
          CHECK_LT(std::fabs(ll[i] - pybeagle[i]), 0.00011);
          CHECK_LT(std::fabs(gradients[i].log_likelihood_ - pybeagle[i]), 0.00011);
        }
        auto last = gradients.back().gradient_.at("branch_lengths");
        std::sort(last.begin(), last.end());
        CHECK_LT(std::fabs(last.front() - -904.18956), 0.0001);
        CHECK_LT(std::fabs(last.back() - 2296.55028), 0.0001);
        CHECK_LT(std::fabs(last[26]), 1e-300);
        CHECK_LT(std::fabs(last[27]), 1e-300);
      }
    }
  }
  {  // one handle, several shards (thread_count FatBeagles in the reference): trees dealt in
     // contiguous blocks give bit-identical per-tree results; empty collections and trees
     // on the wrong taxon count are handled as the reference does / refused
    auto trees = TreeCollection::ParseNexusFile(data + "/DS1.subsampled_10.t");
    SitePattern pattern(Alignment::ReadFasta(data + "/DS1.fasta"), trees.taxon_names_);
    const PhyloModelSpecification weib{"JC69", "weibull+4", "strict"};
    Engine one(EngineSpecification{1, {}, true}, weib, pattern);
    Engine two(EngineSpecification{2, {}, true, {0, 0}}, weib, pattern);
    Engine three(EngineSpecification{3, {}, true, {0, 0, 0}}, weib, pattern);
    ParamMatrix params(trees.TreeCount(), one.ParameterCount());
    params.SetBlock(0, 1, {0.8});
    const auto a = one.Gradients(trees.trees_, params, false);
    for (Engine* e : {&two, &three}) {
      const auto b = e->Gradients(trees.trees_, params, false);
      CHECK_LT(a.size() == b.size() ? 0 : 1, 1);
      for (size_t i = 0; i < a.size() && i < b.size(); i++) {
        CHECK_LT(a[i].log_likelihood_ == b[i].log_likelihood_ ? 0 : 1, 1);
        CHECK_LT(a[i].gradient_.at("branch_lengths") == b[i].gradient_.at("branch_lengths") ? 0 : 1, 1);
        CHECK_LT(a[i].gradient_.at("site_model") == b[i].gradient_.at("site_model") ? 0 : 1, 1);
      }
    }
    CHECK_LT(one.LogLikelihoods(UnrootedTreeCollection{}, ParamMatrix(0, one.ParameterCount()), false).size(), 1);
    CHECK_LT(two.Gradients(UnrootedTreeCollection{}, ParamMatrix(0, one.ParameterCount()), false).size(), 1);
    bool threw = false;
    try {
      UnrootedTreeCollection bad(1, trees.trees_[0]);
      bad[0].parent_ids.pop_back();
      one.LogLikelihoods(bad, ParamMatrix(1, one.ParameterCount()), false);
    } catch (const std::runtime_error&) {
      threw = true;
    }
    CHECK_LT(threw ? 0 : 1, 1);
  }
  {  // fluA rooted: log-likelihood with Jacobian, ratio gradient ends
    auto parsed = TreeCollection::ParseNewickFile(data + "/fluA.tree");
    SitePattern pattern(Alignment::ReadFasta(data + "/fluA.fa"), parsed.taxon_names_);
    RootedTreeCollection trees;
    for (const auto& t : parsed.trees_) {
      RootedFlatTree rt;
      rt.parent_ids = t.parent_ids;
      rt.branch_lengths = t.branch_lengths;
      rt.SetTipDates(ParseDatesFromTaxonNames(parsed.taxon_names_));
      rt.InitializeTimeTreeUsingBranchLengths();
      rt.rates_.assign(rt.rates_.size(), 0.001);
      trees.push_back(rt);
    }
    Engine engine(EngineSpecification{1, {}, true}, simple, pattern);
    ParamMatrix params(1, engine.ParameterCount());
    const auto ll = engine.LogLikelihoods(trees, params, false);
    CHECK_LT(std::fabs(ll[0] - (-4777.616349 + -9.25135166)), 0.0001);
    const auto g = engine.Gradients(trees, params, false);
    CHECK_LT(std::fabs(g[0].log_likelihood_ - -4777.616349), 0.0001);
    const auto& ratios = g[0].gradient_.at("ratios_root_height");
    CHECK_LT(std::fabs(ratios.front() - -0.593654), 0.0001);
    CHECK_LT(std::fabs(ratios.back() - 19.936861), 0.0001);
    bool threw = false;
    try {
      RootedTreeCollection bad = trees;
      bad[0].height_ratios_.clear();
      engine.Gradients(bad, params, false);
    } catch (const std::runtime_error&) {
      threw = true;
    }
    CHECK_LT(threw ? 0 : 1, 1);
  }
  std::printf(failures ? "FAILED (%d)\n" : "engine_example: all checks passed\n", failures);
  return failures ? 1 : 0;
}
