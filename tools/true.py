#!/usr/bin/env python3
"""Stands in for check_agpr_file.py when a NEGATIVE-CONTROL library is built on purpose (-DDG16_ACC_CLOBBER_R5)."""
