#!/usr/bin/env python3
"""Two-second GPU sanity of the widen path: register the keys of a small batch, verify it with 8-bit combs, with explicit 16-bit wide
combs (self-checked against the host builder) and under the default width policy; the three bitmaps must equal the generator's."""
import sys, os, ctypes, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
import consensus_amd as sbv, synth
sbv.init(0)
n = 4096
t, valid = synth.gen_batch(0x77, n, 5, 8)
t2 = t.reshape(n, 160)
keys = [bytes(k) for k in np.unique(t2[:, 96:160], axis=0)]
reg = sbv.register_keys(keys)
slot_of = dict(zip(keys, reg))
slots = [slot_of[bytes(t2[i, 96:160])] for i in range(n)]
rsh = bytes(np.ascontiguousarray(t2[:, :96]).reshape(-1))
narrow = sbv.verify_batch_keyed(rsh, slots, n)
sbv.wide_keys(16, 64)
sbv.widen_keys(reg)
ok16 = sbv.wide_selfcheck(reg[0])
wide16 = sbv.verify_batch_keyed(rsh, slots, n)
sbv.wide_keys()            # auto: rebuilt at 20 bits
st = sbv.wide_key_stats()
wide20 = sbv.verify_batch_keyed(rsh, slots, n)
print(json.dumps({"selfcheck16": ok16, "stats": st, "equal": narrow == wide16 == wide20 == bytes(valid)}))
