# API level of the reference this package is a drop-in for (schpf/_version.py of scHPF 0.5.0);
# stored in model files as `version`.
__version__ = '0.5.0'
