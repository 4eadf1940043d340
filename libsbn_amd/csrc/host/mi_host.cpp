// See mi_host.hpp.  Our own implementation of the reference's input formats and
// id conventions; checked bit-exactly against fixtures produced by the reference's
// own code (tests/test_host_formats.py).
#include "mi_host.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <functional>
#include <memory>
#include <regex>
#include <sstream>

namespace mihost {

void Failwith(const std::string& message) { throw std::runtime_error(message); }

// ======================================================================
// Alignment
// ======================================================================
size_t Alignment::Length() const {
  if (data_.empty()) Failwith("Must have sequences in an alignment to ask for a Length.");
  return data_.begin()->second.size();
}

bool Alignment::IsValid() const {
  if (data_.empty()) return false;
  const size_t len = Length();
  for (const auto& kv : data_)
    if (kv.second.size() != len) return false;
  return true;
}

const std::string& Alignment::at(const std::string& taxon) const {
  auto it = data_.find(taxon);
  if (it == data_.end()) Failwith("Taxon '" + taxon + "' not found in alignment.");
  return it->second;
}

Alignment Alignment::ReadFasta(const std::string& fname) {
  std::ifstream in(fname);
  if (!in.good()) Failwith("Could not open '" + fname + "'");
  std::unordered_map<std::string, std::string> data;
  std::string line, name, seq;
  auto flush = [&]() {
    if (name.empty()) return;
    if (!data.emplace(name, seq).second) Failwith("Failed to insert: taxon '" + name + "' repeated");
  };
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    if (line[0] == '>') {
      flush();
      name = line.substr(1);
      seq.clear();
    } else {
      seq += line;
    }
  }
  flush();
  Alignment alignment(std::move(data));
  if (!alignment.IsValid()) Failwith("Sequences of the alignment are not all the same length.");
  return alignment;
}

// ======================================================================
// SitePattern
// ======================================================================
int SitePattern::SymbolCode(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    // gaps and, "for now", every degenerate nucleotide code (reference issue #162)
    case '-': case 'N': case 'X': case '?': case 'B': case 'D': case 'H': case 'K':
    case 'M': case 'R': case 'S': case 'U': case 'V': case 'W': case 'Y':
      return 4;
    default: {
      char msg[50];
      std::snprintf(msg, sizeof msg, "Symbol '%c' not known.", c);
      Failwith(msg);
    }
  }
}

int SitePattern::AminoAcidCode(char c) {
  static const char order[] = "ARNDCQEGHILKMFPSTWYV";
  const char u = (c >= 'a' && c <= 'z') ? (char)(c - 'a' + 'A') : c;
  for (int i = 0; i < 20; i++)
    if (order[i] == u) return i;
  switch (u) {  // gap, unknown, ambiguity codes, stop, rare residues: no information
    case '-': case 'X': case '?': case 'B': case 'Z': case 'J': case '*': case '.':
    case 'U': case 'O':
      return 20;
    default: {
      char msg[50];
      std::snprintf(msg, sizeof msg, "Symbol '%c' not known.", c);
      Failwith(msg);
    }
  }
}

namespace {
// The pattern order the reference exposes is the iteration order of a
// std::unordered_map keyed by the column with this (boost hash_combine style) hash
// returning int; using the same container and hash reproduces it exactly
// (site_pattern.cpp:67-75,80,105-114).
struct ColumnHash {
  int operator()(const std::vector<int>& column) const {
    int h = column[0];
    for (size_t i = 1; i < column.size(); i++)
      h ^= column[i] + 0x9e3779b9 + (h << 6) + (h >> 2);
    return h;
  }
};
}  // namespace

SitePattern::SitePattern(const Alignment& alignment, const std::vector<std::string>& taxon_names,
                         bool protein) {
  const size_t n = taxon_names.size();
  if (n != alignment.SequenceCount())
    Failwith("The number of taxa in the trees and in the alignment differ.");
  std::vector<const std::string*> rows(n);
  for (size_t i = 0; i < n; i++) rows[i] = &alignment.at(taxon_names[i]);
  site_count_ = alignment.Length();
  state_count_ = protein ? 20 : 4;
  std::unordered_map<std::vector<int>, double, ColumnHash> seen;
  std::vector<int> column(n);
  for (size_t pos = 0; pos < site_count_; pos++) {
    for (size_t i = 0; i < n; i++)
      column[i] = protein ? AminoAcidCode((*rows[i])[pos]) : SymbolCode((*rows[i])[pos]);
    auto it = seen.find(column);
    if (it == seen.end()) seen.insert({column, 1.});
    else it->second += 1.;
  }
  patterns_.assign(n, {});
  for (size_t i = 0; i < n; i++) patterns_[i].reserve(seen.size());
  weights_.reserve(seen.size());
  for (const auto& kv : seen) {
    for (size_t i = 0; i < n; i++) patterns_[i].push_back(kv.first[i]);
    weights_.push_back(kv.second);
  }
}

std::vector<int32_t> SitePattern::FlatPatterns() const {
  std::vector<int32_t> flat;
  flat.reserve(SequenceCount() * PatternCount());
  for (const auto& row : patterns_) flat.insert(flat.end(), row.begin(), row.end());
  return flat;
}

// ======================================================================
// Newick / Nexus
// ======================================================================
namespace {

struct RawNode {
  int leaf_id = -1;  // >= 0 for leaves
  std::vector<std::unique_ptr<RawNode>> children;
  bool has_length = false;
  double length = 0.;
  int max_leaf = -1;
  int id = -1;
};

struct TaxonTable {
  std::map<std::string, uint32_t> ids;  // name as written -> leaf id
  bool complete = false;
  uint32_t next_id = 0;
};

// Tokens per src/scanner.ll:52-81: punctuation ( ) , : ; | LABEL = graphic chars
// other than ( ) ; , : ' [ ] | QUOTED = ('...')+ | [&...] comments.
class NewickParser {
 public:
  NewickParser(const std::string& text, TaxonTable& taxa) : s_(text), taxa_(taxa) {}

  std::unique_ptr<RawNode> ParseTree() {
    auto root = ParseFancyNode();
    SkipBlank();
    if (pos_ >= s_.size() || s_[pos_] != ';') Fail("expected ';'");
    pos_++;
    taxa_.complete = true;
    return root;
  }

 private:
  const std::string& s_;
  TaxonTable& taxa_;
  size_t pos_ = 0;

  [[noreturn]] void Fail(const std::string& what) {
    Failwith("Newick parse error at column " + std::to_string(pos_ + 1) + ": " + what);
  }
  void SkipBlank() {
    while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\r' ||
                                s_[pos_] == '\n'))
      pos_++;
  }
  static bool IsLabelChar(char c) {
    return std::isgraph(static_cast<unsigned char>(c)) && std::string("();,:'[]").find(c) ==
                                                              std::string::npos;
  }
  bool SkipComment() {  // [& ... ]
    SkipBlank();
    if (pos_ + 1 < s_.size() && s_[pos_] == '[' && s_[pos_ + 1] == '&') {
      const size_t close = s_.find(']', pos_);
      if (close == std::string::npos) Fail("unterminated [& comment");
      pos_ = close + 1;
      return true;
    }
    return false;
  }
  std::string ReadLabel() {
    const size_t start = pos_;
    while (pos_ < s_.size() && IsLabelChar(s_[pos_])) pos_++;
    if (pos_ == start) Fail("expected a label");
    return s_.substr(start, pos_ - start);
  }
  std::string ReadQuoted() {  // ('[^']*')+ kept verbatim, dequoted later
    const size_t start = pos_;
    while (pos_ < s_.size() && s_[pos_] == '\'') {
      const size_t close = s_.find('\'', pos_ + 1);
      if (close == std::string::npos) Fail("unterminated quote");
      pos_ = close + 1;
    }
    return s_.substr(start, pos_ - start);
  }

  std::unique_ptr<RawNode> ParseNode() {
    SkipBlank();
    if (pos_ >= s_.size()) Fail("unexpected end of input");
    auto node = std::make_unique<RawNode>();
    if (s_[pos_] == '(') {
      pos_++;
      for (;;) {
        node->children.push_back(ParseFancyNode());
        SkipBlank();
        if (pos_ < s_.size() && s_[pos_] == ',') {
          pos_++;
          continue;
        }
        break;
      }
      SkipBlank();
      if (pos_ >= s_.size() || s_[pos_] != ')') Fail("expected ')'");
      pos_++;
      // labels on internal nodes carry no information for us
      SkipBlank();
      if (pos_ < s_.size() && IsLabelChar(s_[pos_])) ReadLabel();
      else if (pos_ < s_.size() && s_[pos_] == '\'') ReadQuoted();
      SkipComment();
      if (node->children.empty()) Fail("empty node list");
    } else {
      const std::string name = s_[pos_] == '\'' ? ReadQuoted() : ReadLabel();
      SkipComment();
      if (!taxa_.complete) {  // parser.yy:95-101: first tree defines the numbering
        if (!taxa_.ids.emplace(name, taxa_.next_id).second)
          Failwith("Taxon '" + name + "' appears twice in the first tree.");
        node->leaf_id = static_cast<int>(taxa_.next_id++);
      } else {
        auto it = taxa_.ids.find(name);
        if (it == taxa_.ids.end())
          Failwith("Taxon '" + name + "' is not known in our taxon set.\n" +
                   "Either it is missing in the translate block or it didn't appear in the "
                   "first tree.");
        node->leaf_id = static_cast<int>(it->second);
      }
    }
    return node;
  }

  std::unique_ptr<RawNode> ParseFancyNode() {  // node [":" [&comment]? number]
    auto node = ParseNode();
    SkipBlank();
    if (pos_ < s_.size() && s_[pos_] == ':') {
      pos_++;
      SkipComment();
      SkipBlank();
      const std::string number = ReadLabel();
      try {
        size_t used = 0;
        node->length = std::stod(number, &used);
        node->has_length = true;
      } catch (...) {
        Failwith("Float conversion failed on branch length '" + number + "'");
      }
      SkipComment();
    }
    return node;
  }
};

int ComputeMaxLeafAndSort(RawNode* node) {
  if (node->leaf_id >= 0) return node->max_leaf = node->leaf_id;
  for (auto& child : node->children) ComputeMaxLeafAndSort(child.get());
  std::sort(node->children.begin(), node->children.end(),
            [](const auto& a, const auto& b) { return a->max_leaf < b->max_leaf; });
  for (size_t i = 1; i < node->children.size(); i++)
    if (node->children[i]->max_leaf == node->children[i - 1]->max_leaf)
      Failwith("Tie observed between subtrees.\nDo you have a taxon name repeated?");
  return node->max_leaf = node->children.back()->max_leaf;
}

// node.cpp:341-357 (Polish) + tree.cpp:14-28: ids and branch lengths by id.
FlatTree Flatten(RawNode* root, size_t leaf_count) {
  ComputeMaxLeafAndSort(root);
  if (static_cast<size_t>(root->max_leaf) + 1 != leaf_count)
    Failwith("Tree does not contain every taxon of the collection.");
  int next_id = static_cast<int>(leaf_count);
  std::vector<std::pair<int, int>> edges;  // child id, parent id
  std::vector<std::pair<int, double>> lengths;
  std::function<void(RawNode*)> assign = [&](RawNode* node) {
    for (auto& child : node->children) assign(child.get());
    node->id = node->leaf_id >= 0 ? node->leaf_id : next_id++;
    for (auto& child : node->children) edges.push_back({child->id, node->id});
    if (node->has_length) lengths.push_back({node->id, node->length});
  };
  assign(root);
  FlatTree tree;
  tree.parent_ids.assign(root->id, -1);
  tree.branch_lengths.assign(root->id + 1, 0.);
  for (const auto& e : edges) tree.parent_ids[e.first] = e.second;
  for (const auto& l : lengths) tree.branch_lengths[l.first] = l.second;
  for (int p : tree.parent_ids)
    if (p < 0) Failwith("Leaf ids of the tree are not contiguous.");
  return tree;
}

std::vector<std::string> NamesById(const TaxonTable& taxa) {
  std::vector<std::string> names(taxa.ids.size());
  for (const auto& kv : taxa.ids) names[kv.second] = DequoteString(kv.first);
  return names;
}

// driver.cpp:37-57: one tree per line; anything before the first '(' is dropped.
void ParseNewickLines(std::istream& in, TaxonTable& taxa, std::vector<FlatTree>& trees,
                      std::vector<std::unique_ptr<RawNode>>* keep = nullptr) {
  std::string line;
  std::vector<std::unique_ptr<RawNode>> raw;
  while (std::getline(in, line)) {
    const size_t start = line.find_first_of('(');
    if (line.empty() || start == std::string::npos) continue;
    line.erase(0, start);
    NewickParser parser(line, taxa);
    raw.push_back(parser.ParseTree());
  }
  const size_t leaf_count = taxa.ids.size();
  for (auto& r : raw) trees.push_back(Flatten(r.get(), leaf_count));
  if (keep) *keep = std::move(raw);
}

}  // namespace

std::string DequoteString(const std::string& s) {
  if (s.empty()) return s;
  const char delimiter = s[0];
  if (delimiter != '\'' && delimiter != '"') return s;
  std::string out;
  for (size_t i = 1; i < s.size(); i++) {
    if (s[i] == '\\' && i + 1 < s.size()) {
      out += s[++i];
    } else if (s[i] == delimiter) {
      break;
    } else {
      out += s[i];
    }
  }
  return out;
}

TreeCollection TreeCollection::ParseNewickFile(const std::string& fname) {
  std::ifstream in(fname.c_str());
  if (!in) Failwith("Cannot open the File : " + fname);
  TaxonTable taxa;
  TreeCollection collection;
  ParseNewickLines(in, taxa, collection.trees_);
  collection.taxon_names_ = NamesById(taxa);
  return collection;
}

TreeCollection TreeCollection::ParseNewickString(const std::string& newick) {
  std::istringstream in(newick);
  TaxonTable taxa;
  TreeCollection collection;
  ParseNewickLines(in, taxa, collection.trees_);
  if (collection.trees_.empty()) Failwith("No tree found in the Newick string.");
  collection.taxon_names_ = NamesById(taxa);
  return collection;
}

TreeCollection TreeCollection::ParseNexusFile(const std::string& fname) {
  try {
    std::ifstream in(fname.c_str());
    if (!in) throw std::runtime_error("Cannot open file.");
    std::string line;
    std::getline(in, line);
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line != "#NEXUS") throw std::runtime_error("Putative Nexus file doesn't begin with #NEXUS.");
    auto lower_line = [&]() {
      std::getline(in, line);
      if (!line.empty() && line.back() == '\r') line.pop_back();
      std::transform(line.begin(), line.end(), line.begin(),
                     [](unsigned char c) { return std::tolower(c); });
    };
    do {
      if (in.eof()) throw std::runtime_error("Finished reading and couldn't find 'begin trees;'");
      lower_line();
    } while (line != "begin trees;");
    lower_line();
    size_t first = line.find_first_not_of(" \t");
    if (first == std::string::npos || line.compare(first, 9, "translate") != 0 ||
        line.find_first_not_of(" \t", first + 9) != std::string::npos)
      throw std::runtime_error("Missing translate block.");
    // translate entries: "<digits><ws><long name>[,;]"; the k-th entry is leaf k
    // (driver.cpp:105-117); a lone ';' or the first non-entry line ends the block.
    TaxonTable taxa;
    std::vector<std::string> long_names;
    std::streampos body = in.tellg();
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      size_t p = line.find_first_not_of(" \t");
      if (p == std::string::npos) break;
      size_t digits_end = p;
      while (digits_end < line.size() && std::isdigit(static_cast<unsigned char>(line[digits_end])))
        digits_end++;
      if (digits_end == p || digits_end >= line.size() ||
          !std::isspace(static_cast<unsigned char>(line[digits_end])))
        break;  // includes the lone ';' line
      std::string name = line.substr(digits_end + 1);
      if (!name.empty() && (name.back() == ',' || name.back() == ';')) name.pop_back();
      if (name.find_first_of(",;") != std::string::npos) break;
      if (!taxa.ids.emplace(line.substr(p, digits_end - p), taxa.next_id).second)
        throw std::runtime_error("Repeated short name in translate block.");
      taxa.next_id++;
      long_names.push_back(DequoteString(name));
      body = in.tellg();
    }
    if (long_names.empty()) throw std::runtime_error("No taxa found in translate block!");
    taxa.complete = true;
    in.clear();
    in.seekg(body);
    TreeCollection collection;
    ParseNewickLines(in, taxa, collection.trees_);
    collection.taxon_names_ = long_names;
    return collection;
  } catch (const std::exception& exception) {
    Failwith("Problem parsing '" + fname + "':\n" + exception.what());
  }
}

// ======================================================================
// Flat trees
// ======================================================================
std::vector<std::vector<int32_t>> ChildrenOf(const FlatTree& tree, size_t leaf_count) {
  const int nodes = static_cast<int>(tree.NodeCount());
  const int n = static_cast<int>(leaf_count);
  std::vector<int> maxleaf(nodes, -1);
  for (int v = 0; v < n; v++) maxleaf[v] = v;
  std::vector<std::vector<int32_t>> children(nodes);
  for (int v = 0; v < nodes - 1; v++) {
    const int p = tree.parent_ids[v];
    if (p <= v || p >= nodes || p < n) Failwith("parent id vector is not in post-order id form");
    maxleaf[p] = std::max(maxleaf[p], maxleaf[v]);
  }
  for (int v = 0; v < nodes - 1; v++) children[tree.parent_ids[v]].push_back(v);
  for (auto& c : children)
    std::sort(c.begin(), c.end(), [&](int a, int b) { return maxleaf[a] < maxleaf[b]; });
  return children;
}

size_t FlatTree::RootChildCount() const {
  const int root = static_cast<int>(NodeCount()) - 1;
  return std::count(parent_ids.begin(), parent_ids.end(), root);
}

// ======================================================================
// Rooted time trees
// ======================================================================
void RootedFlatTree::SetTipDates(const std::vector<double>& dates) {
  const size_t N = NodeCount(), n = (N + 1) / 2;
  if (RootChildCount() != 2)
    Failwith("Failed to create a RootedTree out of a topology that isn't bifurcating at the "
             "root. Perhaps you are trying to parse unrooted trees into a RootedSBNInstance?");
  if (dates.size() != n) Failwith("Wrong size vector in TagDateMapOfDateVector");
  node_heights_.assign(N, 0.);
  rates_.assign(N - 1, 1.0);
  rate_count_ = 1;
  node_bounds_.assign(N, 0.);
  for (size_t i = 0; i < n; i++) node_bounds_[i] = node_heights_[i] = dates[i];
  const auto children = ChildrenOf(*this, n);
  for (size_t v = n; v < N; v++)
    node_bounds_[v] = std::max(node_bounds_[children[v][0]], node_bounds_[children[v][1]]);
}

void RootedFlatTree::InitializeTimeTreeUsingBranchLengths() {
  if (!TipDatesHaveBeenSet())
    Failwith("Attempted access of a time tree member that requires the tip dates to be set. "
             "Have you set dates for your time trees?");
  const size_t N = NodeCount(), n = (N + 1) / 2, root = N - 1;
  const auto children = ChildrenOf(*this, n);
  height_ratios_.assign(n - 1, 0.);
  for (size_t v = n; v < N; v++) {
    const int c0 = children[v][0], c1 = children[v][1];
    node_heights_[v] = node_heights_[c0] + branch_lengths[c0];
    const double diff = std::fabs(node_heights_[c1] + branch_lengths[c1] - node_heights_[v]);
    if (diff > 1e-4)
      Failwith("Tree isn't time-calibrated in RootedTree::InitializeTimeTreeUsingBranchLengths. "
               "Height difference: " + std::to_string(diff));
  }
  height_ratios_[root - n] = node_heights_[root];
  for (size_t v = n; v < root; v++)
    height_ratios_[v - n] = (node_heights_[v] - node_bounds_[v]) /
                            (node_heights_[parent_ids[v]] - node_bounds_[v]);
}

void RootedFlatTree::InitializeTimeTreeUsingHeightRatios(const std::vector<double>& ratios) {
  if (!TipDatesHaveBeenSet())
    Failwith("Attempted access of a time tree member that requires the tip dates to be set. "
             "Have you set dates for your time trees?");
  const size_t N = NodeCount(), n = (N + 1) / 2, root = N - 1;
  if (ratios.size() != n - 1) Failwith("Wrong number of height ratios.");
  height_ratios_ = ratios;
  node_heights_[root] = ratios[root - n];
  for (size_t i = root; i-- > 0;) {  // parents have larger ids: a pre-order
    const size_t parent = parent_ids[i];
    if (i >= n)
      node_heights_[i] =
          node_bounds_[i] + ratios[i - n] * (node_heights_[parent] - node_bounds_[i]);
    branch_lengths[i] = node_heights_[parent] - node_heights_[i];
  }
}

std::vector<double> ParseDatesFromTaxonNames(const std::vector<std::string>& names) {
  static const std::regex date_regex(R"raw(^.+_(\d*\.?\d+(?:[eE][-+]?\d+)?)$)raw");
  std::vector<double> dates;
  double max_date = -INFINITY;
  for (const auto& name : names) {
    std::smatch m;
    if (!std::regex_match(name, m, date_regex)) Failwith("Couldn't parse a date from:" + name);
    dates.push_back(std::stod(m[1].str()));
    max_date = std::max(max_date, dates.back());
  }
  for (auto& d : dates) d = max_date - d;
  return dates;
}

}  // namespace mihost
