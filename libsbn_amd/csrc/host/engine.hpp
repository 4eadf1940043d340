// C++ adapter with the public shape of libsbn's Engine (src/engine.hpp:26-54) on
// top of the C ABI (include/mi_phylo.h).  Header-only; links against
// libmi_phylo.so (+ libmi_phylo_host.so for the tree / site-pattern types).
//
//   reference                               here
//   EngineSpecification  engine.hpp:20-24   mihost::EngineSpecification
//   PhyloModelSpecification phylo_model.hpp:13-17   mihost::PhyloModelSpecification
//   PhyloGradient        tree_gradient.hpp:10-19    mihost::PhyloGradient
//   {Unrooted,Rooted}TreeCollection         std::vector<FlatTree> / <RootedFlatTree>
//   EigenMatrixXdRef (row-major)            mihost::ParamMatrix
// Errors are std::runtime_error, like Failwith (src/sugar.hpp:67-78).
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../../include/mi_phylo.h"
#include "mi_host.hpp"

namespace mihost {

struct PhyloModelSpecification {
  std::string substitution_, site_, clock_;
};

struct EngineSpecification {
  // The reference makes thread_count_ FatBeagle instances and deals the trees of a call to
  // them (engine.cpp:23-27, fat_beagle.hpp:119-149).  Here the executors are GPUs: the trees
  // of a call are dealt, in contiguous blocks, to min(thread_count_, visible devices) devices
  // driven by ONE handle (mi_engine_create_sharded); on a one-GPU machine any thread count
  // gives one engine.  device_shards_ overrides that (testing: several logical shards).
  size_t thread_count_;
  std::vector<int> beagle_flag_vector_;  // accepted and ignored
  bool use_tip_states_;
  std::vector<int32_t> device_shards_ = {};
};

using GradientMap = std::map<std::string, std::vector<double>>;
struct PhyloGradient {
  double log_likelihood_ = 0.;
  GradientMap gradient_;
};

struct ParamMatrix {  // row-major [rows x cols]
  size_t rows = 0, cols = 0;
  std::vector<double> data;
  ParamMatrix() = default;
  ParamMatrix(size_t r, size_t c) : rows(r), cols(c), data(r * c, 0.) {}
  double& operator()(size_t r, size_t c) { return data[r * cols + c]; }
  void SetBlock(size_t start, size_t length, const std::vector<double>& values) {
    for (size_t r = 0; r < rows; r++)
      for (size_t i = 0; i < length; i++) data[r * cols + start + i] = values[i % values.size()];
  }
};

using UnrootedTreeCollection = std::vector<FlatTree>;
using RootedTreeCollection = std::vector<RootedFlatTree>;
using BlockSpecificationMap = std::map<std::string, std::pair<size_t, size_t>>;

class Engine {
 public:
  Engine(const EngineSpecification& engine_specification,
         const PhyloModelSpecification& specification, SitePattern site_pattern)
      : site_pattern_(std::move(site_pattern)) {
    if (engine_specification.thread_count_ == 0)
      Failwith("Thread count needs to be strictly positive.");
    mi_engine_spec spec{};
    spec.taxon_count = static_cast<int32_t>(site_pattern_.SequenceCount());
    spec.pattern_count = static_cast<int32_t>(site_pattern_.PatternCount());
    spec.state_count = 4;
    if (specification.substitution_ == "JC69") spec.subst_model = MI_SUBST_JC69;
    else if (specification.substitution_ == "GTR") spec.subst_model = MI_SUBST_GTR;
    else if (specification.substitution_ == "WAG") {  // 20 states, built-in table
      spec.subst_model = MI_SUBST_REVERSIBLE;
      spec.state_count = 20;
    } else Failwith("Substitution model not known: " + specification.substitution_);
    if (site_pattern_.StateCount() != spec.state_count)
      Failwith("The site pattern is coded in a " + std::to_string(site_pattern_.StateCount()) +
               "-state alphabet, the substitution model " + specification.substitution_ +
               " has " + std::to_string(spec.state_count) + " states.");
    if (specification.site_ == "constant") {
      spec.site_model = MI_SITE_CONSTANT;
      spec.category_count = 1;
    } else if (specification.site_.rfind("weibull", 0) == 0) {  // site_model.cpp:15-22
      spec.site_model = MI_SITE_WEIBULL;
      const auto plus = specification.site_.find("+");
      spec.category_count =
          plus == std::string::npos ? 4 : std::stoi(specification.site_.substr(plus + 1));
    } else {
      Failwith("Site model not known: " + specification.site_);
    }
    if (specification.clock_ == "none") spec.clock_model = MI_CLOCK_NONE;
    else if (specification.clock_ == "strict") spec.clock_model = MI_CLOCK_STRICT;
    else Failwith("Clock model not known: " + specification.clock_);
    spec.use_tip_states = engine_specification.use_tip_states_;
    spec.device = -1;
    category_count_ = spec.category_count;
    is_gtr_ = spec.subst_model == MI_SUBST_GTR;
    const auto tips = site_pattern_.FlatPatterns();
    std::vector<int32_t> shards = engine_specification.device_shards_;
    if (shards.empty()) {
      // thread_count executors (engine.cpp:23-27) = that many devices, counted from the
      // caller's CURRENT HIP device (ordinal -1 - i, include/mi_phylo.h): one executor stays
      // on the device the process selected -- under one-process-per-GPU launches every rank
      // would otherwise pile onto device 0
      const size_t devices = static_cast<size_t>(std::max(1, mi_device_count()));
      for (size_t i = 0; i < std::min(engine_specification.thread_count_, devices); i++)
        shards.push_back(-1 - static_cast<int32_t>(i));
    }
    taxon_count_ = static_cast<size_t>(spec.taxon_count);
    Check(mi_engine_create_sharded(&spec, static_cast<int32_t>(shards.size()), shards.data(),
                                   MI_SHARD_TREES, nullptr, nullptr, tips.data(), nullptr,
                                   site_pattern_.GetWeights().data(), &handle_));
    for (int i = 0; i < mi_engine_block_count(handle_); i++) {
      const char* name;
      int32_t start, length;
      Check(mi_engine_block(handle_, i, &name, &start, &length));
      block_specification_[name] = {static_cast<size_t>(start), static_cast<size_t>(length)};
    }
  }
  ~Engine() { mi_engine_destroy(handle_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  const BlockSpecificationMap& GetPhyloModelBlockSpecification() const {
    return block_specification_;
  }
  size_t ParameterCount() const { return block_specification_.at("entire").second; }

  std::vector<double> LogLikelihoods(const UnrootedTreeCollection& trees,
                                     const ParamMatrix& params, const bool rescaling) const {
    std::vector<int32_t> parents;
    std::vector<double> bl;
    Flatten(trees, params, false, &parents, &bl);
    std::vector<double> out(trees.size());
    if (trees.empty()) return out;  // FatBeagleParallelize returns an empty vector
    Check(mi_engine_log_likelihoods_unrooted(handle_, static_cast<int32_t>(trees.size()),
                                             parents.data(), bl.data(), params.data.data(),
                                             rescaling, out.data()));
    return out;
  }

  std::vector<double> LogLikelihoods(const RootedTreeCollection& trees,
                                     const ParamMatrix& params, const bool rescaling) const {
    return RootedLogLikelihoods(trees, params, rescaling, true);
  }

  std::vector<double> UnrootedLogLikelihoods(const RootedTreeCollection& trees,
                                             const ParamMatrix& params,
                                             const bool rescaling) const {
    return RootedLogLikelihoods(trees, params, rescaling, false);
  }

  std::vector<PhyloGradient> Gradients(const UnrootedTreeCollection& trees,
                                       const ParamMatrix& params, const bool rescaling) const {
    const size_t T = trees.size(), N = 2 * site_pattern_.SequenceCount() - 1;
    // (the flat arrays are the engine's own, re-used from call to call: no allocation, no
    // zero-filling of a megabyte of outputs per call)
    std::vector<int32_t>& parents = scratch_parents_;
    std::vector<double>& bl = scratch_bl_;
    Flatten(trees, params, false, &parents, &bl);
    if (trees.empty()) return {};
    Grow(&scratch_ll_, T);
    Grow(&scratch_a_, T * N);
    Grow(&scratch_site_, T);
    Grow(&scratch_subst_, T * 8);
    const double *ll = scratch_ll_.data(), *g = scratch_a_.data(), *site = scratch_site_.data(),
                 *subst = scratch_subst_.data();
    Check(mi_engine_gradients_unrooted(handle_, static_cast<int32_t>(T), parents.data(),
                                       bl.data(), params.data.data(), rescaling, scratch_ll_.data(),
                                       scratch_a_.data(), category_count_ > 1 ? scratch_site_.data() : nullptr,
                                       is_gtr_ ? scratch_subst_.data() : nullptr));
    std::vector<PhyloGradient> out(T);
    for (size_t t = 0; t < T; t++) {
      out[t].log_likelihood_ = ll[t];
      // (keys in map order, each placed at the end with its vector built in place)
      GradientMap& m = out[t].gradient_;
      m.emplace_hint(m.end(), std::piecewise_construct, std::forward_as_tuple("branch_lengths"),
                     std::forward_as_tuple(g + t * N, g + (t + 1) * N));
      AddModelGradients(&out[t], site[t], subst + 8 * t);
    }
    return out;
  }

  std::vector<PhyloGradient> Gradients(const RootedTreeCollection& trees,
                                       const ParamMatrix& params, const bool rescaling) const {
    const size_t T = trees.size(), n = site_pattern_.SequenceCount(), N = 2 * n - 1;
    std::vector<int32_t> parents, rate_counts;
    std::vector<double> bl, rates, heights, bounds, ratios;
    Flatten(trees, params, true, &parents, &bl);
    if (trees.empty()) return {};
    for (const auto& tree : trees) {
      if (!tree.TimeTreeHasBeenInitialized())  // rooted_tree.hpp:50-53
        Failwith("Attempted access of a time tree member that requires the time tree to be "
                 "initialized. Have you set dates for your time trees, and initialized the "
                 "time trees?");
      rates.insert(rates.end(), tree.rates_.begin(), tree.rates_.end());
      rate_counts.push_back(static_cast<int32_t>(tree.rate_count_));
      heights.insert(heights.end(), tree.node_heights_.begin(), tree.node_heights_.end());
      bounds.insert(bounds.end(), tree.node_bounds_.begin(), tree.node_bounds_.end());
      ratios.insert(ratios.end(), tree.height_ratios_.begin(), tree.height_ratios_.end());
    }
    std::vector<double> ll(T), gr(T * (n - 1)), gc(T * (N - 1)), site(T), subst(T * 8);
    Check(mi_engine_gradients_rooted(handle_, static_cast<int32_t>(T), parents.data(), bl.data(),
                                     params.data.data(), rates.data(), rate_counts.data(),
                                     heights.data(), bounds.data(), ratios.data(), rescaling,
                                     ll.data(), gr.data(), gc.data(), site.data(),
                                     subst.data()));
    std::vector<PhyloGradient> out(T);
    for (size_t t = 0; t < T; t++) {
      out[t].log_likelihood_ = ll[t];
      out[t].gradient_["ratios_root_height"].assign(gr.begin() + t * (n - 1),
                                                    gr.begin() + (t + 1) * (n - 1));
      const size_t clock_len = trees[t].rate_count_ == 1 ? 1 : N - 1;
      out[t].gradient_["clock_model"].assign(gc.begin() + t * (N - 1),
                                             gc.begin() + t * (N - 1) + clock_len);
      AddModelGradients(&out[t], site[t], &subst[8 * t]);
    }
    return out;
  }

 private:
  SitePattern site_pattern_;
  mi_engine* handle_ = nullptr;
  BlockSpecificationMap block_specification_;
  int category_count_ = 1;
  bool is_gtr_ = false;
  size_t taxon_count_ = 0;
  // flat arrays of the last call (an Engine is not used from two threads at once: neither is
  // the C handle behind it)
  mutable std::vector<int32_t> scratch_parents_;
  mutable std::vector<double> scratch_bl_, scratch_ll_, scratch_a_, scratch_site_, scratch_subst_;

  static void Check(int rc) {
    if (rc != 0) Failwith(mi_last_error());
  }

  template <class TColl>
  void Flatten(const TColl& trees, const ParamMatrix& params, bool rooted,
               std::vector<int32_t>* parents, std::vector<double>* bl) const {
    if (trees.size() != params.rows)  // fat_beagle.hpp:138
      Failwith("We param_matrix needs as many rows as we have trees.");
    if (params.cols != ParameterCount()) Failwith("Parameters are the wrong dimension!");
    // every tree must be on the alignment's taxa: the C ABI reads fixed-size rows
    const size_t n = taxon_count_;
    const size_t want_parents = rooted ? 2 * n - 2 : 2 * n - 3, want_bl = want_parents + 1;
    for (const auto& tree : trees)
      if (tree.parent_ids.size() != want_parents || tree.branch_lengths.size() != want_bl)
        Failwith("Tree does not have the taxon count of the site pattern (expected " +
                 std::to_string(want_parents) + " parent ids and " + std::to_string(want_bl) +
                 " branch lengths).");
    parents->resize(trees.size() * want_parents);
    bl->resize(trees.size() * want_bl);
    int32_t* pp = parents->data();
    double* pb = bl->data();
    for (const auto& tree : trees) {
      pp = std::copy(tree.parent_ids.begin(), tree.parent_ids.end(), pp);
      pb = std::copy(tree.branch_lengths.begin(), tree.branch_lengths.end(), pb);
    }
  }

  std::vector<double> RootedLogLikelihoods(const RootedTreeCollection& trees,
                                           const ParamMatrix& params, bool rescaling,
                                           bool with_jacobian) const {
    std::vector<int32_t> parents;
    std::vector<double> bl, rates, heights, bounds;
    Flatten(trees, params, true, &parents, &bl);
    if (trees.empty()) return {};
    if (with_jacobian)
      for (const auto& tree : trees) {
        if (!tree.TimeTreeHasBeenInitialized())
          Failwith("Attempted access of a time tree member that requires the time tree to be "
                   "initialized. Have you set dates for your time trees, and initialized the "
                   "time trees?");
        rates.insert(rates.end(), tree.rates_.begin(), tree.rates_.end());
        heights.insert(heights.end(), tree.node_heights_.begin(), tree.node_heights_.end());
        bounds.insert(bounds.end(), tree.node_bounds_.begin(), tree.node_bounds_.end());
      }
    std::vector<double> out(trees.size());
    Check(mi_engine_log_likelihoods_rooted(
        handle_, static_cast<int32_t>(trees.size()), parents.data(), bl.data(),
        params.data.data(), with_jacobian ? rates.data() : nullptr,
        with_jacobian ? heights.data() : nullptr, with_jacobian ? bounds.data() : nullptr,
        with_jacobian, rescaling, out.data()));
    return out;
  }

  // ("site_model" < "substitution_model", both after the other keys: hinted at the end)
  void AddModelGradients(PhyloGradient* g, double site, const double* subst) const {
    GradientMap& m = g->gradient_;
    if (category_count_ > 1)
      m.emplace_hint(m.end(), std::piecewise_construct, std::forward_as_tuple("site_model"),
                     std::forward_as_tuple(size_t{1}, site));
    if (is_gtr_)
      m.emplace_hint(m.end(), std::piecewise_construct, std::forward_as_tuple("substitution_model"),
                     std::forward_as_tuple(subst, subst + 8));
  }
  static void Grow(std::vector<double>* v, size_t count) {
    if (v->size() < count) v->resize(count);
  }
};

}  // namespace mihost
