from .sublayers import Sampler, MultiHeadAttention, FeedForward
from .layers import EncoderLayer, DecoderLayer
from .modules import (Embeddings, PositionalEncoding, Norm, get_clones, get_src_mask, get_trg_mask,
                      nopeak_mask)
from .vaetf import Vaetf
from .cvaetf import Cvaetf
from .build_model import get_model, load_checkpoint, load_state, model_dict
from .forward_propagation1 import forward_propagation
