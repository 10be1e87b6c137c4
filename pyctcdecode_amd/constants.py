"""Decoding defaults; values must equal the reference's (pyctcdecode/constants.py:5-18)."""
import math

DEFAULT_ALPHA = 0.5
DEFAULT_BETA = 1.5
DEFAULT_UNK_LOGP_OFFSET = -10.0
DEFAULT_BEAM_WIDTH = 100
DEFAULT_HOTWORD_WEIGHT = 10.0
DEFAULT_PRUNE_LOGP = -10.0
DEFAULT_PRUNE_BEAMS = False
DEFAULT_MIN_TOKEN_LOGP = -5.0
DEFAULT_SCORE_LM_BOUNDARY = True

AVG_TOKEN_LEN = 6  # expected characters per word; scales the partial-word OOV penalty
MIN_TOKEN_CLIP_P = 1e-15  # probability floor applied to every frame
LOG_BASE_CHANGE_FACTOR = 1.0 / math.log10(math.e)  # log10 -> ln
