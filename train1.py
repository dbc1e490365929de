#!/usr/bin/env python3
"""Drop-in replacement for the reference's train1.py (same single-dash flags, same
model_folder outputs: records.log, train_/valid_{epoch}.csv, model_{epoch}.pt) running the
MI355X-native engine.  See gct_plus_amd/train1.py."""
from gct_plus_amd.train1 import cli

if __name__ == "__main__":
    cli()
