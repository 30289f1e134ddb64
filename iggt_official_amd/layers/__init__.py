from .blocks import Attention, Block, LayerScale, Mlp, NestedTensorBlock, MemEffAttention  # noqa: F401
from .patch_embed import PatchEmbed  # noqa: F401
from .rope import PositionGetter, RotaryPositionEmbedding2D  # noqa: F401
