"""Build the C part of the oracle (test infrastructure) with gcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'gist_oracle.c')
OUT = os.path.join(HERE, 'libgist_oracle.so')


def build(force=False):
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    cmd = ['gcc', '-O3', '-march=x86-64-v2', '-fopenmp', '-shared', '-fPIC',
           '-o', OUT, SRC]
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
