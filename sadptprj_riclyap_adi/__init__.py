"""Drop-in package name the reference imports
(``/root/reference/optcont_main.py:13-14``, ``/root/reference/solve_dae_ric.py:3-4``).

With the repository root on ``sys.path``,
``import sadptprj_riclyap_adi.proj_ric_utils as pru`` and
``import sadptprj_riclyap_adi.lin_alg_utils as lau`` resolve to the MI355X
implementation in :mod:`optconpy_amd`.
"""
