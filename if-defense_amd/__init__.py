"""if-defense_amd: MI355X-native ConvONet-Opt restoration (IF-Defense hot path).

Import name: ``ifdefense_amd`` (see the shim at the repo root).  Importing the package does not load
the HIP library; constructing a ``Restorer`` (or calling ``load_library``) does, and fails loudly when
``csrc/libifd.so`` is absent - there is no CPU fallback.
"""
from . import weights            # noqa: F401
from ._lib import LIB_PATH, load as load_library   # noqa: F401
from .runtime import IfdError, OnetRestorer, Restorer, planes_from_channel_last, planes_to_channel_last   # noqa: F401
from .pipeline import (DefenseArgs, defend_npz_test_data, defend_npz_train_test_data, defend_point_cloud,   # noqa: F401
                       defend_stream, get_save_name, remesh_point_cloud)

__all__ = ["Restorer", "OnetRestorer", "IfdError", "weights", "load_library", "LIB_PATH", "planes_to_channel_last",
           "planes_from_channel_last", "DefenseArgs", "defend_point_cloud", "defend_npz_test_data",
           "defend_npz_train_test_data", "defend_stream", "get_save_name", "remesh_point_cloud"]
