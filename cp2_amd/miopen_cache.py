"""MIOpen's solver search, done once and shipped.

The encoders' convolutions run in MIOpen (north_star keeps them there).  With `torch.backends.cudnn.benchmark = True` MIOpen
benchmarks every applicable solver the first time it meets a convolution problem and keeps the ranking in its USER find
database (`$MIOPEN_USER_DB_PATH`, default ~/.config/miopen): on a fresh machine that search is 60 s of the first training
step per process (measured: a 30-step `bench.py` run takes 80 s of wall clock with an empty database, 21 s with a filled one),
W ranks on one host run it concurrently against the same files, and its single-shot timings settle on different solver
sets from box to box (+-2 % of throughput).  `cp2_amd/miopen_db/` holds the user find / performance databases MIOpen itself
wrote on an MI355X for the convolution problems of BASELINE configs 2, 4 and 5 (text files named after the MIOpen build
and the device, so another build or GPU simply ignores them and searches as before).  `use_shipped_find_db()` copies them
into a per-user cache directory -- MIOpen appends whatever it still has to search there, the repository stays untouched --
and points MIOpen at it, unless the caller already chose a database directory.

On by default (CP2_MIOPEN_DB=0 = search as stock PyTorch does).  Same-box A/B after a warm-up run, alternating, 40 steps
each: 2564 / 2562 / 2561 img/s from the shipped rankings, 2547 / 2572 / 2556 from a fresh search -- the same speed (four
independent searches pick the same solver for 79 of the 80 problems of the config-2 step, and the shipped files agree with
them); what changes is the start-up: 21-27 s of wall clock for a 26-step bench run instead of 80-83 s.  (A first comparison
had the shipped run 2 % behind: it was the first process on a cold box both times -- run order, not rankings.)
"""
from __future__ import annotations

import os
import shutil
from typing import Optional

SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_find_db() -> Optional[str]:
    """Call before the first convolution runs.  Returns the directory MIOpen was pointed at, or None when nothing was done
    (CP2_MIOPEN_DB=0, MIOPEN_USER_DB_PATH already set, no shipped files, or the cache directory cannot be written)."""
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("CP2_MIOPEN_DB", "1") == "0" or not os.path.isdir(SHIPPED):
        return None
    files = [f for f in os.listdir(SHIPPED) if f.endswith((".udb.txt", ".ufdb.txt"))]
    if not files:
        return None
    dst = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "cp2_amd", "miopen_db")
    try:
        os.makedirs(dst, exist_ok=True)
        for f in files:
            out = os.path.join(dst, f)
            if not os.path.exists(out):                    # keep what earlier runs added; ranks racing here copy the same bytes
                tmp = f"{out}.{os.getpid()}.tmp"
                shutil.copyfile(os.path.join(SHIPPED, f), tmp)
                os.replace(tmp, out)
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
