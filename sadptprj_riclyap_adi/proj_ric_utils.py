"""``sadptprj_riclyap_adi.proj_ric_utils`` -> :mod:`optconpy_amd.proj_ric_utils`."""
from optconpy_amd.proj_ric_utils import *  # noqa: F401,F403
from optconpy_amd.proj_ric_utils import __all__  # noqa: F401
