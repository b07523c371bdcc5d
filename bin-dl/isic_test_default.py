#!/usr/bin/env python3
"""isic test script (default) -- same flags as the reference's bin-dl/isic_test_default.py, running on librcu_hip."""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    try:
        parser = argparse.ArgumentParser(description='isic test script (default)')
        parser.add_argument('-config_file', type=str, help='the json file name containing the train configuration')
        parser.add_argument('-config_id', type=str, help='the id of a known config (is ignored when config_file set)')
        args = parser.parse_args()
        from rcu_amd import scripts
        scripts.test_default('isic', args.config_file, args.config_id)
    finally:
        logging.exception('')  # log the exception
