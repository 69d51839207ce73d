#!/usr/bin/env python3
"""Drop-in command line for dna-factory's population generator on MI355X.

    python pop_factory.py -s <cases> -c <controls> -x <snps> [-p deleterious.yml] [-f 0.005] [-m 0.5] [-z 6] ...

Same flags and output files as the reference's pop_factory.py; the implementation is dna-factory_amd/cli.py.
"""
import sys

from dna_factory_amd.cli import main, parse_cmd_args  # noqa: F401

if __name__ == "__main__":
    main(sys.argv[1:])
