"""petsc_amd -- MI355X-native FEM global-assembly engine (drop-in for the assembly hot path of
the CuiCu-618/petsc drivers).  The product is petsc_amd/csrc/libfemasm.so (include/femasm.h);
petsc_amd.femasm is its ctypes binding."""
from . import femasm  # noqa: F401
