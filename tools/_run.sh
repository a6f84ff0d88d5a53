#!/bin/bash
for i in 1 2 3; do python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1; done
