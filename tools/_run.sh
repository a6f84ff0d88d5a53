#!/bin/bash
python tools/_tiler_stencil.py 2>&1 | tail -7
