"""pcdet/utils/spconv_utils.py: the backbone imports `spconv` and `replace_feature` from here."""
from ..ops import spconv  # noqa: F401


def replace_feature(out, new_features):
    out.features = new_features
    return out
