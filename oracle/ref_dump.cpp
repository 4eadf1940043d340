// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// ref_dump: a small driver (our own code) around the *unmodified* structural
// sources of the reference (the "integer half" of the hot path: alignment,
// site-pattern compression, Newick/Nexus parsing, node-id assignment,
// Detrifurcate, traversal orders).  It is compiled by oracle/Makefile directly
// from /root/reference/src/*.cpp into oracle/_ref/ref_dump and is used only to
//   (1) generate the committed fixtures under tests/golden/ and
//   (2) validate oracle/phylo_oracle.c's restatement of the same logic.
// The floating-point half of the reference (fat_beagle.cpp -> BEAGLE, Eigen)
// cannot be built in this image and is NOT stubbed; see DESIGN.md.
//
// Build note: every public Driver::Parse* entry calls
// TaxonNameMunging::DequoteTagStringMap, whose TU needs Eigen (absent).  We
// therefore call the private Driver::ParseNewick(std::ifstream&) directly (Nexus
// files: our own translate-block reader feeds Driver::taxa_ first);
// the file is compiled with -fno-access-control and linked with --gc-sections
// so the unreferenced public wrappers are dropped.  No stand-in code exists.
//
// Reference entry points exercised (file:line in /root/reference/src):
//   Alignment::ReadFasta             alignment.cpp:40-72
//   SitePattern::SitePattern/Compress site_pattern.hpp:18-22, site_pattern.cpp:77-115
//   Driver::ParseNewick              driver.cpp:37-57
//   Tree::Tree(topology, TagDoubleMap) tree.cpp:14-28 (Node::Polish node.cpp:341-357)
//   UnrootedTree::Detrifurcate       unrooted_tree.cpp:27-37
//   Node::BinaryIdPostorder / TripleIdPreorderBifurcating  node.cpp:190-261
//   Node::ParentIdVector             node.hpp:153

#include <cstdio>
#include <fstream>
#include <iostream>
#include <cctype>
#include <string>
#include <vector>

#include "alignment.hpp"
#include "driver.hpp"
#include "site_pattern.hpp"
#include "unrooted_tree.hpp"

static void PrintIntVec(const char* key, const std::vector<int>& v, bool last = false) {
  std::printf("\"%s\": [", key);
  for (size_t i = 0; i < v.size(); i++) std::printf("%s%d", i ? "," : "", v[i]);
  std::printf("]%s", last ? "" : ", ");
}

static void PrintDoubleVec(const char* key, const std::vector<double>& v,
                           bool last = false) {
  std::printf("\"%s\": [", key);
  for (size_t i = 0; i < v.size(); i++) std::printf("%s%.17g", i ? "," : "", v[i]);
  std::printf("]%s", last ? "" : ", ");
}

static std::vector<int> ParentIds(const Node::NodePtr& topology) {
  std::vector<int> out;
  for (auto x : topology->ParentIdVector()) out.push_back(static_cast<int>(x));
  return out;
}

// Nexus front end (ours): the k-th entry of the translate block gets leaf id k
// and trees refer to taxa by the short token -- the rule stated at
// driver.cpp:105-117.  We register short tokens in Driver::taxa_ and then let the
// reference's own ParseNewick consume the remaining "tree ... = (...);" lines.
static TreeCollection ParseNexusTrees(Driver& driver, const std::string& fname,
                                      TagStringMap& long_name_taxon_map) {
  driver.Clear();
  std::ifstream in(fname.c_str());
  if (!in) Failwith("Cannot open file " + fname);
  std::string line;
  bool in_translate = false;
  uint32_t leaf_id = 0;
  std::streampos body_start = 0;
  while (std::getline(in, line)) {
    std::string low = line;
    for (auto& c : low) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    const auto first = low.find_first_not_of(" \t");
    if (!in_translate) {
      if (first != std::string::npos && low.compare(first, 9, "translate") == 0) {
        in_translate = true;
        body_start = in.tellg();
      }
      continue;
    }
    // inside the translate block: "<short> <long>[,;]" or a lone ";"
    if (first == std::string::npos) continue;
    if (line[first] == ';') break;
    if (!std::isdigit(static_cast<unsigned char>(line[first]))) break;  // first tree line
    const auto short_end = line.find_first_of(" \t", first);
    const std::string short_name = line.substr(first, short_end - first);
    auto long_begin = line.find_first_not_of(" \t", short_end);
    auto long_end = line.find_last_not_of(" \t\r");
    bool closes = false;
    if (line[long_end] == ',' || line[long_end] == ';') {
      closes = (line[long_end] == ';');
      long_end--;
    }
    const std::string long_name = line.substr(long_begin, long_end - long_begin + 1);
    SafeInsert(driver.taxa_, short_name, leaf_id);
    SafeInsert(long_name_taxon_map, PackInts(leaf_id, 1), long_name);
    leaf_id++;
    body_start = in.tellg();
    if (closes) break;
  }
  if (leaf_id == 0) Failwith("No taxa found in translate block of " + fname);
  driver.taxa_complete_ = true;
  in.clear();
  in.seekg(body_start);
  return driver.ParseNewick(in);
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr,
                 "usage: ref_dump <fasta> <treefile> <newick|nexus> <unrooted|rooted>\n");
    return 2;
  }
  const std::string fasta = argv[1], treefile = argv[2], format = argv[3],
                    kind = argv[4];
  Driver driver;
  TagStringMap tag_taxon_map;
  Tree::TreeVector trees;
  if (format == "nexus") {
    auto coll = ParseNexusTrees(driver, treefile, tag_taxon_map);
    trees = coll.Trees();
  } else {
    driver.Clear();
    std::ifstream in(treefile.c_str());
    if (!in) Failwith("Cannot open " + treefile);
    auto coll = driver.ParseNewick(in);
    trees = coll.Trees();
    tag_taxon_map = coll.TagTaxonMap();
  }
  const size_t n = tag_taxon_map.size();
  std::vector<std::string> names(n);
  for (const auto& [tag, name] : tag_taxon_map) names[MaxLeafIDOfTag(tag)] = name;

  auto alignment = Alignment::ReadFasta(fasta);
  SitePattern site_pattern(alignment, tag_taxon_map);

  std::printf("{\"taxon_count\": %zu, \"site_count\": %zu, \"pattern_count\": %zu, ", n,
              site_pattern.SiteCount(), site_pattern.PatternCount());
  std::printf("\"taxon_names\": [");
  for (size_t i = 0; i < n; i++) std::printf("%s\"%s\"", i ? "," : "", names[i].c_str());
  std::printf("], ");
  std::printf("\"patterns\": [");
  for (size_t i = 0; i < n; i++) {
    std::printf("%s[", i ? "," : "");
    const auto& row = site_pattern.GetPatterns()[i];
    for (size_t p = 0; p < row.size(); p++) std::printf("%s%d", p ? "," : "", row[p]);
    std::printf("]");
  }
  std::printf("], ");
  PrintDoubleVec("weights", site_pattern.GetWeights());

  std::printf("\"trees\": [");
  for (size_t t = 0; t < trees.size(); t++) {
    const Tree& tree = trees[t];
    std::printf("%s{", t ? "," : "");
    PrintIntVec("parent_ids", ParentIds(tree.Topology()));
    PrintDoubleVec("branch_lengths", tree.BranchLengths());
    Tree bif = tree;
    if (kind == "unrooted") {
      UnrootedTree unrooted(tree.Topology(), tree.BranchLengths());
      bif = unrooted.Detrifurcate();
      PrintIntVec("bif_parent_ids", ParentIds(bif.Topology()));
      PrintDoubleVec("bif_branch_lengths", bif.BranchLengths());
    }
    std::vector<int> post, pre;
    bif.Topology()->BinaryIdPostorder([&post](int a, int b, int c) {
      post.push_back(a);
      post.push_back(b);
      post.push_back(c);
    });
    bif.Topology()->TripleIdPreorderBifurcating([&pre](int a, int b, int c) {
      pre.push_back(a);
      pre.push_back(b);
      pre.push_back(c);
    });
    PrintIntVec("postorder_triples", post);
    PrintIntVec("preorder_triples", pre, true);
    std::printf("}");
  }
  std::printf("]}\n");
  return 0;
}
