"""`python -m inquistr_amd call|combine|outlier ...` — forwards to the native CLI (inquistr_amd/lib/inquistr), which is
the program a user of the reference would run in place of `inquiSTR call` / `inquiSTR combine` / `inquiSTR outlier`."""
import os
import subprocess
import sys

from .call import CLI_PATH


def main() -> int:
    if not os.path.exists(CLI_PATH):
        print(f"{CLI_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` first", file=sys.stderr)
        return 1
    return subprocess.call([CLI_PATH] + sys.argv[1:])


if __name__ == "__main__":
    sys.exit(main())
