#!/usr/bin/env python
"""``python tools/selfcheck.py [IN.wav] [--seeded] [--batch 32]`` == ``python -m voicefixer_amd --selfcheck ...``: default (Winograd) vs
direct vs bf16x3 arithmetic on the loaded checkpoint, stage by stage (voicefixer_amd/selfcheck.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_amd import selfcheck  # noqa: E402

if __name__ == "__main__":
    sys.exit(selfcheck.main())
