// Host-side C++ of the MI355X engine: the data formats either side of the hot
// path, re-stated (not copied) from the reference so that a libsbn user finds the
// same behaviour:
//   Alignment::ReadFasta                 src/alignment.cpp:40-72
//   SitePattern (compression + order)    src/site_pattern.cpp:16-131
//   Newick / Nexus ingest, leaf numbering  src/driver.cpp:37-146, src/parser.yy:76-112,
//                                          src/scanner.ll:52-81
//   Node ids (Polish), child order       src/node.cpp:32-59,341-357
//   Tree(topology, tag->length)          src/tree.cpp:14-28
//   RootedTree time-tree state           src/rooted_tree.cpp:20-102
//   dates from taxon names               src/taxon_name_munging.cpp:46-78
// Trees are kept in the flat form the C ABI consumes (parent-id vectors).
// Plain C++17, no HIP: builds and is tested on CPU.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace mihost {

// src/sugar.hpp:67-78 (Failwith): every error is a std::runtime_error.
[[noreturn]] void Failwith(const std::string& message);

// ---- alignment + site patterns -------------------------------------------
class Alignment {
 public:
  Alignment() = default;
  explicit Alignment(std::unordered_map<std::string, std::string> data) : data_(std::move(data)) {}
  static Alignment ReadFasta(const std::string& fname);
  size_t SequenceCount() const { return data_.size(); }
  size_t Length() const;
  bool IsValid() const;
  const std::string& at(const std::string& taxon) const;
  const std::unordered_map<std::string, std::string>& Data() const { return data_; }

 private:
  std::unordered_map<std::string, std::string> data_;
};

class SitePattern {
 public:
  SitePattern() = default;
  // taxon_names[i] = name of leaf id i (the reference passes a tag->taxon map).
  // protein = false: the reference's DNA symbol table (site_pattern.cpp:16-46);
  // true: the 20 amino acids in the order ARNDCQEGHILKMFPSTWYV -> 0..19, gaps and ambiguity
  // codes -> 20 (not in the reference: the alphabet of the 20-state engine).
  SitePattern(const Alignment& alignment, const std::vector<std::string>& taxon_names,
              bool protein = false);
  const std::vector<std::vector<int>>& GetPatterns() const { return patterns_; }
  const std::vector<double>& GetWeights() const { return weights_; }
  size_t PatternCount() const { return patterns_.empty() ? 0 : patterns_[0].size(); }
  size_t SequenceCount() const { return patterns_.size(); }
  size_t SiteCount() const { return site_count_; }
  // states of the alphabet the patterns are coded in (4: DNA, 20: amino acids); an Engine
  // whose model has another state count refuses the pattern (the codes would be read as the
  // other alphabet's: DNA's gap code 4 is Cys in amino-acid order)
  int StateCount() const { return state_count_; }
  // row-major [taxon][pattern] copy for the C ABI
  std::vector<int32_t> FlatPatterns() const;
  static int SymbolCode(char c);  // site_pattern.cpp:16-56
  static int AminoAcidCode(char c);

 private:
  std::vector<std::vector<int>> patterns_;
  std::vector<double> weights_;
  size_t site_count_ = 0;
  int state_count_ = 4;
};

// ---- trees ---------------------------------------------------------------------
// One tree in the reference's id convention: leaves 0..n-1, internal nodes in
// post-order, root last.  parent_ids has one entry per non-root node,
// branch_lengths one per node (root entry included, as Tree::branch_lengths_).
struct FlatTree {
  std::vector<int32_t> parent_ids;
  std::vector<double> branch_lengths;
  size_t NodeCount() const { return branch_lengths.size(); }
  size_t RootChildCount() const;
};

struct RootedFlatTree : FlatTree {
  // RootedTree state (src/rooted_tree.hpp:90-104); empty until initialised.
  std::vector<double> node_bounds_, height_ratios_, node_heights_, rates_;
  size_t rate_count_ = 0;
  bool TipDatesHaveBeenSet() const { return !node_bounds_.empty(); }
  bool TimeTreeHasBeenInitialized() const { return !height_ratios_.empty(); }
  void SetTipDates(const std::vector<double>& dates_by_leaf);   // rooted_tree.cpp:20-43
  void InitializeTimeTreeUsingBranchLengths();                  // rooted_tree.cpp:45-81
  void InitializeTimeTreeUsingHeightRatios(const std::vector<double>& ratios);  // :83-102
};

class TreeCollection {
 public:
  std::vector<FlatTree> trees_;
  std::vector<std::string> taxon_names_;  // by leaf id
  size_t TreeCount() const { return trees_.size(); }
  size_t TaxonCount() const { return taxon_names_.size(); }
  // Driver::ParseNewickFile / ParseNexusFile / ParseString (src/driver.cpp:59-165)
  static TreeCollection ParseNewickFile(const std::string& fname);
  static TreeCollection ParseNexusFile(const std::string& fname);
  static TreeCollection ParseNewickString(const std::string& newick);
};

// src/taxon_name_munging.cpp:46-78: trailing "_<number>", then max - date.
std::vector<double> ParseDatesFromTaxonNames(const std::vector<std::string>& names);
std::string DequoteString(const std::string& s);  // taxon_name_munging.cpp:17-29

// children (ordered by max leaf id) of every node of a tree in id convention;
// throws on vectors that are not in that convention.
std::vector<std::vector<int32_t>> ChildrenOf(const FlatTree& tree, size_t leaf_count);

}  // namespace mihost
