#!/usr/bin/env python3
"""isic test script (auxiliary segm.) -- same flags as the reference's bin-dl/isic_test_auxiliary_segm.py, running on librcu_hip."""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    try:
        parser = argparse.ArgumentParser(description='isic test script (auxiliary segm.)')
        parser.add_argument('-config_file', type=str, help='the json file name containing the train configuration')

        args = parser.parse_args()
        from rcu_amd import scripts
        scripts.test_auxiliary_segm('isic', args.config_file)
    finally:
        logging.exception('')  # log the exception
