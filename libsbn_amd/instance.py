"""The part of pylibsbn's instance surface that sits on the Engine path
(src/pylibsbn.cpp:192-201,249-254,274-284,354-359), with the same method names and
argument meaning, so that the reference's own tests read the same here:

    inst = libsbn_amd.unrooted_instance("charlie")
    inst.read_newick_file("data/hello.nwk")
    inst.read_fasta_file("data/hello.fasta")
    inst.prepare_for_phylo_likelihood(PhyloModelSpecification("JC69","constant","strict"), 2)
    inst.log_likelihoods()          # -> numpy vector
    inst.phylo_gradients()          # -> [PhyloGradient]
    inst.get_phylo_model_param_block_map()["Weibull shape"][:] = 0.1

Everything SBN-related (supports, sampling, training) is outside the hot path and
not provided.
"""
import numpy as np

from . import _hostapi
from .engine import Engine, PhyloModelSpecification  # noqa: F401


class _Instance:
    _rooted = False

    def __init__(self, name=""):
        self.name = name
        self.tree_collection = None
        self._fasta = None
        self._engine = None
        self._rescaling = False
        self._params = None

    # ---- I/O (generic_sbn_instance.hpp:272-300) ----
    def read_newick_file(self, path):
        self.tree_collection = _hostapi.TreeCollection.of_newick_file(path)
        self._check_rooting()

    def read_nexus_file(self, path):
        self.tree_collection = _hostapi.TreeCollection.of_nexus_file(path)
        self._check_rooting()

    def read_fasta_file(self, path):
        self._fasta = path

    def tree_count(self):
        return 0 if self.tree_collection is None else self.tree_collection.tree_count()

    def taxon_names(self):
        return list(self.tree_collection.taxon_names)

    def _check_rooting(self):
        n = self.tree_collection.taxon_count()
        want = 2 * n - 1 if self._rooted else 2 * n - 2
        for bl in self.tree_collection.branch_lengths:
            if len(bl) != want:
                raise RuntimeError(
                    "Failed to create a RootedTree out of a topology that isn't bifurcating at "
                    "the root. Perhaps you are trying to parse unrooted trees into a "
                    "RootedSBNInstance?" if self._rooted else
                    "Expected a tree with a trifucation at the root in the constructor of "
                    "UnrootedTree.")

    # ---- engine (generic_sbn_instance.hpp:247-270,303-309) ----
    def prepare_for_phylo_likelihood(self, model_specification, thread_count,
                                     beagle_flags=(), use_tip_states=True,
                                     tree_count_option=None):
        if self._fasta is None or self.tree_count() == 0:
            raise RuntimeError("Load an alignment and trees into your instance before "
                               "preparing for phylogenetic likelihood calculation.")
        # (a 20-state model reads the alignment with the amino-acid alphabet)
        protein = model_specification.substitution in ("WAG", "reversible")
        pats, w, _ = self.tree_collection.site_pattern(self._fasta, protein=protein)
        if self._engine is not None:
            self._engine.close()
        self._engine = Engine(model_specification, pats, w, use_tip_states=use_tip_states,
                              thread_count=thread_count)  # beagle_flags: accepted, ignored
        self.resize_phylo_model_params(tree_count_option)

    def resize_phylo_model_params(self, tree_count_option=None):
        count = tree_count_option if tree_count_option else self.tree_count()
        if count == 0:
            raise RuntimeError("Please add trees to your instance by sampling or loading "
                               "before preparing for phylogenetic likelihood calculation.")
        # the reference leaves the matrix uninitialised (Eigen resize); zeros here
        self._params = np.zeros((count, self._engine.param_count))

    def get_engine(self):
        if self._engine is None:
            raise RuntimeError("Engine not available. Call PrepareForPhyloLikelihood to make "
                               "an engine for phylogenetic likelihood computation "
                               "computation.")
        return self._engine

    def get_phylo_model_params(self):
        return self._params

    def get_phylo_model_param_block_map(self):
        spec = self.get_engine().block_specification()
        return {k: self._params[:, s:s + l] for k, (s, l) in spec.items()}

    def set_rescaling(self, use_rescaling):
        self._rescaling = bool(use_rescaling)

    def _trees(self):
        tc = self.tree_collection
        if len(self._params) != tc.tree_count():  # fat_beagle.hpp:138
            raise RuntimeError("We param_matrix needs as many rows as we have trees.")
        return np.stack(tc.parent_ids), np.stack(tc.branch_lengths)


class unrooted_instance(_Instance):
    """UnrootedSBNInstance, Engine path only (src/unrooted_sbn_instance.cpp:94-100)."""

    def log_likelihoods(self):
        pid, bl = self._trees()
        return self.get_engine().log_likelihoods(pid, bl, self._params, self._rescaling)

    def phylo_gradients(self, gradient_blocks=None):
        """pylibsbn's phylo_gradients(); gradient_blocks (an extension, see
        Engine.gradients) names the blocks the caller will read."""
        pid, bl = self._trees()
        return self.get_engine().gradients(pid, bl, self._params, self._rescaling,
                                           gradient_blocks=gradient_blocks)


class rooted_instance(_Instance):
    """RootedSBNInstance, Engine path only (src/rooted_sbn_instance.cpp:42-53)."""
    _rooted = True

    def __init__(self, name=""):
        super().__init__(name)
        self.rates, self.rate_counts = None, None
        self.node_heights = self.node_bounds = self.height_ratios = None
        self.tip_dates = None

    def _set_dates(self, dates, initialize_time_trees):
        tc = self.tree_collection
        n = tc.taxon_count()
        self.tip_dates = np.asarray(dates, float)
        T = tc.tree_count()
        self.rates = np.ones((T, 2 * n - 2))       # RootedTree::SetTipDates: strict clock, 1.0
        self.rate_counts = np.ones(T, np.int32)
        if initialize_time_trees:
            hs, bs, rs = [], [], []
            for pid, bl in zip(tc.parent_ids, tc.branch_lengths):
                h, b, r = _hostapi.time_tree_from_branch_lengths(pid, bl, self.tip_dates)
                hs.append(h), bs.append(b), rs.append(r)
            self.node_heights, self.node_bounds, self.height_ratios = map(np.stack, (hs, bs, rs))

    def parse_dates_from_taxon_names(self, initialize_time_trees=True):
        self._set_dates(self.tree_collection.dates_from_taxon_names(), initialize_time_trees)

    def set_dates_to_be_constant(self, initialize_time_trees=True):
        self._set_dates(np.zeros(self.tree_collection.taxon_count()), initialize_time_trees)

    def _need_time_trees(self):
        if self.node_heights is None:
            raise RuntimeError(
                "Attempted access of a time tree member that requires the time tree to be "
                "initialized. Have you set dates for your time trees, and initialized the time "
                "trees?")

    def log_likelihoods(self):
        self._need_time_trees()
        pid, bl = self._trees()
        return self.get_engine().rooted_log_likelihoods(
            pid, bl, self._params, self.rates, self.node_heights, self.node_bounds,
            self._rescaling, with_jacobian=True)

    def unrooted_log_likelihoods(self):
        pid, bl = self._trees()
        return self.get_engine().rooted_log_likelihoods(pid, bl, self._params,
                                                        rescaling=self._rescaling,
                                                        with_jacobian=False)

    def phylo_gradients(self):
        self._need_time_trees()
        pid, bl = self._trees()
        return self.get_engine().rooted_gradients(
            pid, bl, self._params, self.rates, self.rate_counts, self.node_heights,
            self.node_bounds, self.height_ratios, self._rescaling)
