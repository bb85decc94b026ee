#!/usr/bin/env python3
"""Print the rows of DESIGN.md section 6 from the library's one settings table (csrc/srcnn_settings.hpp, through
srcnn_debug_settings -- no device needed).  tests/test_abi.py checks that DESIGN.md carries exactly these rows."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libsrcnn_amd as S
print("| switch | default; values | effect |\n|---|---|---|")
print(S.debug_settings(markdown=True), end="")
