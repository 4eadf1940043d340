// C API over mi_host.hpp for the Python layer (ctypes) and for tests.
// Error convention: functions return 0 / non-null on success; mih_last_error()
// holds the message otherwise.
#include <cstring>
#include <string>

#include "mi_host.hpp"

using namespace mihost;

namespace {
thread_local std::string g_error;
template <typename F>
auto guarded(F&& f, decltype(f()) on_error) -> decltype(f()) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_error = e.what();
    return on_error;
  }
}
}  // namespace

struct mih_trees {
  TreeCollection collection;
};
struct mih_site_pattern {
  SitePattern pattern;
};

extern "C" {

const char* mih_last_error(void) { return g_error.c_str(); }

mih_trees* mih_parse_newick_file(const char* path) {
  return guarded([&]() { return new mih_trees{TreeCollection::ParseNewickFile(path)}; },
                 (mih_trees*)nullptr);
}
mih_trees* mih_parse_nexus_file(const char* path) {
  return guarded([&]() { return new mih_trees{TreeCollection::ParseNexusFile(path)}; },
                 (mih_trees*)nullptr);
}
mih_trees* mih_parse_newick_string(const char* newick) {
  return guarded([&]() { return new mih_trees{TreeCollection::ParseNewickString(newick)}; },
                 (mih_trees*)nullptr);
}
void mih_trees_free(mih_trees* t) { delete t; }
int32_t mih_tree_count(const mih_trees* t) { return (int32_t)t->collection.TreeCount(); }
int32_t mih_taxon_count(const mih_trees* t) { return (int32_t)t->collection.TaxonCount(); }
const char* mih_taxon_name(const mih_trees* t, int32_t i) {
  return t->collection.taxon_names_.at(i).c_str();
}
int32_t mih_node_count(const mih_trees* t, int32_t tree) {
  return (int32_t)t->collection.trees_.at(tree).NodeCount();
}
int32_t mih_copy_tree(const mih_trees* t, int32_t tree, int32_t* parent_ids,
                      double* branch_lengths) {
  const FlatTree& ft = t->collection.trees_.at(tree);
  std::memcpy(parent_ids, ft.parent_ids.data(), sizeof(int32_t) * ft.parent_ids.size());
  std::memcpy(branch_lengths, ft.branch_lengths.data(), sizeof(double) * ft.branch_lengths.size());
  return 0;
}

mih_site_pattern* mih_site_pattern_from_fasta(const char* fasta, const mih_trees* t) {
  return guarded(
      [&]() {
        return new mih_site_pattern{
            SitePattern(Alignment::ReadFasta(fasta), t->collection.taxon_names_)};
      },
      (mih_site_pattern*)nullptr);
}
mih_site_pattern* mih_site_pattern_from_fasta_protein(const char* fasta, const mih_trees* t) {
  return guarded(
      [&]() {
        return new mih_site_pattern{
            SitePattern(Alignment::ReadFasta(fasta), t->collection.taxon_names_, true)};
      },
      (mih_site_pattern*)nullptr);
}
void mih_site_pattern_free(mih_site_pattern* p) { delete p; }
int32_t mih_pattern_count(const mih_site_pattern* p) { return (int32_t)p->pattern.PatternCount(); }
int32_t mih_site_count(const mih_site_pattern* p) { return (int32_t)p->pattern.SiteCount(); }
int32_t mih_sequence_count(const mih_site_pattern* p) {
  return (int32_t)p->pattern.SequenceCount();
}
int32_t mih_copy_site_pattern(const mih_site_pattern* p, int32_t* patterns, double* weights) {
  const auto flat = p->pattern.FlatPatterns();
  std::memcpy(patterns, flat.data(), sizeof(int32_t) * flat.size());
  std::memcpy(weights, p->pattern.GetWeights().data(),
              sizeof(double) * p->pattern.GetWeights().size());
  return 0;
}

int32_t mih_dates_from_taxon_names(const mih_trees* t, double* out_dates) {
  return guarded(
      [&]() {
        const auto dates = ParseDatesFromTaxonNames(t->collection.taxon_names_);
        std::memcpy(out_dates, dates.data(), sizeof(double) * dates.size());
        return 0;
      },
      1);
}

static RootedFlatTree make_rooted(int32_t n, const int32_t* parent_ids, const double* bl) {
  RootedFlatTree tree;
  tree.parent_ids.assign(parent_ids, parent_ids + 2 * n - 2);
  tree.branch_lengths.assign(bl, bl + 2 * n - 1);
  return tree;
}

int32_t mih_time_tree_from_branch_lengths(int32_t n, const int32_t* parent_ids, const double* bl,
                                          const double* tip_dates, double* node_heights,
                                          double* node_bounds, double* height_ratios) {
  return guarded(
      [&]() {
        RootedFlatTree tree = make_rooted(n, parent_ids, bl);
        tree.SetTipDates(std::vector<double>(tip_dates, tip_dates + n));
        tree.InitializeTimeTreeUsingBranchLengths();
        std::memcpy(node_heights, tree.node_heights_.data(), sizeof(double) * (2 * n - 1));
        std::memcpy(node_bounds, tree.node_bounds_.data(), sizeof(double) * (2 * n - 1));
        std::memcpy(height_ratios, tree.height_ratios_.data(), sizeof(double) * (n - 1));
        return 0;
      },
      1);
}

int32_t mih_time_tree_from_height_ratios(int32_t n, const int32_t* parent_ids,
                                         const double* tip_dates, const double* height_ratios,
                                         double* branch_lengths, double* node_heights,
                                         double* node_bounds) {
  return guarded(
      [&]() {
        std::vector<double> zeros(2 * n - 1, 0.);
        RootedFlatTree tree = make_rooted(n, parent_ids, zeros.data());
        tree.SetTipDates(std::vector<double>(tip_dates, tip_dates + n));
        tree.InitializeTimeTreeUsingHeightRatios(
            std::vector<double>(height_ratios, height_ratios + n - 1));
        std::memcpy(branch_lengths, tree.branch_lengths.data(), sizeof(double) * (2 * n - 1));
        std::memcpy(node_heights, tree.node_heights_.data(), sizeof(double) * (2 * n - 1));
        std::memcpy(node_bounds, tree.node_bounds_.data(), sizeof(double) * (2 * n - 1));
        return 0;
      },
      1);
}

}  // extern "C"
