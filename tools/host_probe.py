#!/usr/bin/env python3
"""prints hpfw_amd.hostinfo's view of the host: usable CPUs and the reference's CPU libraries"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfw_amd import hostinfo  # noqa: E402

print(json.dumps({"cpu": hostinfo.cpu_budget(), "reference_cpu_libraries": hostinfo.reference_cpu_libraries()}))
