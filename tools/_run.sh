#!/bin/bash
python tests/stencil_stress.py 1500 2>&1 | tail -8
