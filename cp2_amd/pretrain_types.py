"""Enum shared by pre-training and fine-tuning checkpoints (reference networks/segment_network.py:14-38);
member names and values are part of the checkpoint contract (main.py:540, segment_network.py:79-92)."""
from enum import Enum


class PretrainType(Enum):
    RANDOM = 0
    NONE = 1
    CP2 = 2
    MIRROR = 3
    BYOL = 4
    MOCO = 5
    PROPOSED = 6
    PIXPRO = 7
    DENSECL_IMGNET = 8
    DINO_IMGNET = 9
    BARLOWTWINS_IMGNET = 10
    VICEREGL_IMGNET = 11
    MOCO_IMGNET = 12
    PIXPRO_IMGNET = 13
    BYOL_IMGNET = 14
    CP2_IMGNET = 15
    MOSREP_IMGNET = 16
    CLOVE_IMGNET = 17
    DENSECL = 18
    PROPOSED_V2 = 19
