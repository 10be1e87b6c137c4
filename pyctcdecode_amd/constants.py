"""Decoding defaults.  Names and values are API: they equal the reference's pyctcdecode/constants.py:5-18
(alpha/beta of the shallow fusion, OOV offset, beam width, hot-word weight, the two pruning
thresholds, history pruning off, boundary scoring on; 6 characters per expected word, the 1e-15
probability floor and ln(10) for kenlm's base-10 scores)."""
import math

(DEFAULT_ALPHA, DEFAULT_BETA, DEFAULT_UNK_LOGP_OFFSET) = (0.5, 1.5, -10.0)
(DEFAULT_BEAM_WIDTH, DEFAULT_HOTWORD_WEIGHT) = (100, 10.0)
(DEFAULT_PRUNE_LOGP, DEFAULT_MIN_TOKEN_LOGP) = (-10.0, -5.0)
(DEFAULT_PRUNE_BEAMS, DEFAULT_SCORE_LM_BOUNDARY) = (False, True)

AVG_TOKEN_LEN = 6
MIN_TOKEN_CLIP_P = 1e-15
LOG_BASE_CHANGE_FACTOR = 1.0 / math.log10(math.e)
