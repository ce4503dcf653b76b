"""The protocol drivers live in the package (mpc_amd/protocols.py); `import protocols` in the tests gives that module itself, so that the
tests run the drivers the benchmark runs, on the GPU backend and on the CPU oracle backend."""
import sys

import mpc_amd.protocols as _real

sys.modules[__name__] = _real
