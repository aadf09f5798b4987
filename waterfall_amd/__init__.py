"""waterfall_amd: MI355X-native DWFST loss / Viterbi hot path behind the reference's own API.

    from waterfall_amd import graphloss, GraphLoss, DenseFsaVec, Fsa, create_fsa_vec
    from waterfall_amd import linear_fsa, remove_epsilon_and_add_self_loops, arc_sort, compose

mirrors what ``waterfall/graph.py`` and the ``ctc_k2`` / ``k2`` branches of ``waterfall/models.py``
import from k2.  All arithmetic runs in libwfdwfst.so (HIP, gfx950).
"""
from .fsa import Fsa, DenseFsaVec, create_fsa_vec
from .graph import GraphLoss, graphloss, graphloss_num_den, tot_scores, best_path, pruned_search, set_output_beam_mode, set_eager_gradient, set_num_den_overlap
from .compile import linear_fsa, remove_epsilon_and_add_self_loops, arc_sort, compose, compile_ctc_k2_graphs
from . import manual_ctc
from . import dist
from . import compile
from .lang import Lang, write_lang_dir
from .decode import (DenseLattice, OneBest, get_lattice, intersect_dense, one_best_decoding, shortest_path,
                     get_aux_labels, linear_fst, connect, invert, compose_words, DenseLatticeW)
from .graph import lattice_words
from .topology import build_topology, topology_text, Topology, TOPOLOGIES

__all__ = [
    "Fsa", "DenseFsaVec", "create_fsa_vec", "GraphLoss", "graphloss", "graphloss_num_den", "tot_scores", "best_path", "pruned_search", "set_output_beam_mode", "set_eager_gradient", "set_num_den_overlap", "manual_ctc", "dist", "Lang", "write_lang_dir", "DenseLattice", "OneBest", "get_lattice", "intersect_dense",
    "one_best_decoding", "shortest_path", "get_aux_labels", "linear_fst", "connect", "invert", "compose_words", "DenseLatticeW", "lattice_words",
    "linear_fsa", "remove_epsilon_and_add_self_loops", "arc_sort", "compose", "compile_ctc_k2_graphs",
    "build_topology", "topology_text", "Topology", "TOPOLOGIES",
]
