#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
