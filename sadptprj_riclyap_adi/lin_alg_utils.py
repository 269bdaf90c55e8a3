"""``sadptprj_riclyap_adi.lin_alg_utils`` -> :mod:`optconpy_amd.lin_alg_utils`."""
from optconpy_amd.lin_alg_utils import *  # noqa: F401,F403
from optconpy_amd.lin_alg_utils import __all__  # noqa: F401
