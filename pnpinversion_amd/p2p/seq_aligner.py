"""API-compatibility alias of the reference module name models/p2p/seq_aligner.py; the implementation is token_align.py."""
from .token_align import (get_mapper, get_refinement_mapper, get_replacement_mapper, get_replacement_mapper_,  # noqa: F401
                          get_word_inds)
