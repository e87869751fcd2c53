"""Device time of the transform alone on the three synthetic kinds (text, DNA, random bytes),
256 MiB blocks unless a size in MiB is given.  Development aid; bench.py is the measurement."""
import sys

import numpy as np

sys.path.insert(0, ".")
from bwtc_amd import hip


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    size = mib << 20
    with hip.Context(0, size) as ctx:
        buf = np.empty(size, np.uint8)
        for kind, name in (("t", "text"), ("d", "dna"), ("r", "random")):
            hip.synth_into(kind, 3, buf)
            best = 1e9
            for _ in range(3):
                ctx.bwt_block(buf, 8)
                st = ctx.stats()
                best = min(best, st.ms_total)
            print("%-7s %d MiB: %.2f ms device, %d rounds" % (name, mib, best, st.rounds), flush=True)


if __name__ == "__main__":
    main()
