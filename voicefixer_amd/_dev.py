"""The ONE development gate of the Python side (the library's own switches are compiled in by `make dev` only, csrc/vfx_common.h).

A normal process consults no ``VFX_*`` environment variable: ``dev_env`` returns the default unless ``VFX_DEV=1`` is set, and the
package reads the environment nowhere else (tests/test_api_cpu.py::test_python_package_reads_the_environment_through_the_dev_gate_only).
With ``VFX_DEV=1`` the development tools (tools/*.py, tools/sanitizer_check.sh) may select another build of the library (``VFX_LIB``),
switch the host FLAC codec off (``VFX_FLAC_NATIVE=0``) or re-run the recorded A/Bs of the engine's launch choices."""
import os


def dev_env(name, default=None):
    if os.environ.get("VFX_DEV") != "1":
        return default
    return os.environ.get(name, default)
