"""Pieces of bench.py that are not the benchmark itself: the synthetic generators (data), engine set-up and the
convergence fits (setup), and the rocprofv3 counter passes behind roofline.traffic (traffic).  bench.py re-exports their
public names; the CPU comparator stays in bench.py (the one place outside tests/ that may call the oracle)."""
