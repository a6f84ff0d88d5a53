#!/bin/bash
python tools/_zn48.py 2>&1 | tail -30
