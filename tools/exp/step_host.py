"""Where a bench step's wall time goes on the host side: the three calls of a step timed from Python, against the library's own phase timers."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import bench as B
import skani_amd as sk
args = B.parse_args([]) if hasattr(B, "parse_args") else None
