#!/bin/bash
python tools/_t48.py
