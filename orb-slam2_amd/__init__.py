"""MI355X-native ORB front end: ORBextractor and the ORBmatcher Hamming kNN over liborbhip.so."""
from . import _lib, sharded, synth  # noqa: F401
from ._lib import OrbhipError  # noqa: F401
from .extractor import KP_DTYPE, ORBextractor  # noqa: F401
from .frame import AREA_QUERY_DTYPE, PROBE_DTYPE, ComputeImageBounds, FrameBatch, UndistortKeyPoints  # noqa: F401
from .matcher import ORBmatcher  # noqa: F401
from .vocabulary import ORBVocabulary  # noqa: F401
