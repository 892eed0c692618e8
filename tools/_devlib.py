"""Imported FIRST by the tools that switch the options of measured-and-rejected experiments (diag_v2 = 1 / 3, sb_qr4, la_split,
band_log2, nb_big, raw_graph_max_n, ...): points FFGP_LIB at the development build (`make -C fidelityfusion_amd/csrc dev` ->
fidelityfusion_amd/libffgp_dev.so) unless the caller chose a library already.  The shipped libffgp.so refuses those keys."""
import os

_dev = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fidelityfusion_amd", "libffgp_dev.so")
if "FFGP_LIB" not in os.environ and os.path.exists(_dev):
    os.environ["FFGP_LIB"] = _dev
