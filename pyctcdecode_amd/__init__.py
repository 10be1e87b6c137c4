"""MI355X-native CTC beam-search decoder with the pyctcdecode API surface
(reference exports: pyctcdecode/__init__.py:2-4)."""
from .alphabet import Alphabet  # noqa: F401
from .decoder import BeamSearchDecoderCTC, build_ctcdecoder  # noqa: F401
from .language_model import LanguageModel  # noqa: F401

__version__ = "0.1.0"
