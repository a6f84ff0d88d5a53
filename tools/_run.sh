#!/bin/bash
python tools/_e2e_stencil.py 2>&1 | tail -10
