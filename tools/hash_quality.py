"""Quality of the n-gram table key (common.h: ngram_key_begin / _push / _end) on a real ARPA file, old chain (splitmix64
finaliser per word, rounds 1-4) against the round-5 mix_step chain: distinct tuples -> distinct 64-bit keys? and how long
are the linear-probing chains at the load factor the host builds its table with (<= 1/4)?
  python tools/hash_quality.py bench_cache/synth_20000w_60000s_o4_s7.arpa [extra random tuples, millions]"""
import sys

import numpy as np

M64 = (1 << 64) - 1


def mix64(x):
    x = x ^ (x >> np.uint64(30))
    x = x * np.uint64(0xBF58476D1CE4E5B9)
    x = x ^ (x >> np.uint64(27))
    x = x * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def push_old(k, ids):
    return mix64(k ^ ids.astype(np.uint64)) + np.uint64(0x13198A2E03707344)


def mix_step(state, x_lo, x_hi):
    a = (state & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    b = (state >> np.uint64(32)).astype(np.uint32)
    a = (a ^ x_lo) * np.uint32(0x9E3779B1)
    a ^= a >> np.uint32(15)
    b = ((b + x_hi) ^ a) * np.uint32(0x85EBCA6B)
    b ^= b >> np.uint32(13)
    return (b.astype(np.uint64) << np.uint64(32)) | a.astype(np.uint64)


def push_new(k, ids):
    ids = ids.astype(np.uint32)
    return mix_step(k, ids, (ids << np.uint32(16)) | (ids >> np.uint32(16)))


def keys_kenlm(tuples):
    """kenlm's own chain (CombineWordHash): the scored word's index, then every context word going back; order in the top byte"""
    A, B = np.uint64(8978948897894561157), np.uint64(17894857484156487943)
    k = tuples[:, -1].astype(np.uint64)
    for j in range(tuples.shape[1] - 2, -1, -1):
        k = (k * A) ^ ((tuples[:, j].astype(np.uint64) + np.uint64(1)) * B)
    k ^= np.uint64(tuples.shape[1]) << np.uint64(56)
    return k


def keys(tuples, push):
    if push is None:
        return keys_kenlm(tuples)
    """tuples: [n, order] word ids, oldest first; key chain newest first"""
    k = np.full(len(tuples), 0x243F6A8885A308D3, dtype=np.uint64)
    for j in range(tuples.shape[1] - 1, -1, -1):
        k = push(k, tuples[:, j])
    k ^= np.uint64(tuples.shape[1]) << np.uint64(56)
    return k


def probes(keys_all):
    size = 16
    while size < 4 * len(keys_all) + 1:
        size <<= 1
    slots = (keys_all & np.uint64(size - 1)).astype(np.int64)
    # linear probing, insertion order: simulate with a python loop over collisions only
    occ = np.zeros(size, dtype=bool)
    total = 0
    worst = 0
    order = np.argsort(slots, kind="stable")
    # (insertion order does not change the total displacement of linear probing)
    cur = -1
    for s in slots[order]:
        p = max(s, cur + 1) if cur >= s else s
        # robin-hood free total displacement == sum over sorted slots of (position - slot) with position = max(slot, prev + 1)
        cur = p
        d = p - s + 1
        total += d
        worst = max(worst, d)
    return size, total / len(keys_all), worst


def main():
    path = sys.argv[1]
    extra = float(sys.argv[2]) if len(sys.argv) > 2 else 0
    vocab = {"<unk>": 0}
    by_order = {}
    section = 0
    with open(path, errors="replace") as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("\\") and line.endswith("-grams:"):
                section = int(line[1:line.index("-")])
                continue
            if not line or line.startswith("\\") or line.startswith("ngram ") or section == 0:
                continue
            parts = line.split("\t")
            toks = parts[1].split(" ")
            if section == 1:
                if toks[0] not in vocab:
                    vocab[toks[0]] = len(vocab)
            else:
                by_order.setdefault(section, []).append([vocab.get(t, 0) for t in toks])
    print("%s: %d words, n-grams per order %s" % (path, len(vocab), {k: len(v) for k, v in by_order.items()}))
    rng = np.random.default_rng(1)
    for name, push in (("old (splitmix64 per word)", push_old), ("round 5 (mix_step per word)", push_new),
                       ("round 6 (kenlm's CombineWordHash)", None)):
        ks = [keys(np.array(v, dtype=np.int64), push) for _, v in sorted(by_order.items())]
        allk = np.concatenate(ks)
        uniq = len(np.unique(allk))
        size, avg, worst = probes(allk)
        print("%-28s %d tuples -> %d distinct keys (%d collisions); table %d slots: %.3f probes per hit on average, longest %d"
              % (name, len(allk), uniq, len(allk) - uniq, size, avg, worst))
        if extra:
            # random + structured tuples far beyond the file: small ids, consecutive ids, permutations of one another
            n = int(extra * 1e6)
            V = len(vocab)
            t = rng.integers(0, V, size=(n, 4))
            t2 = t.copy()
            t2[:, [0, 1]] = t2[:, [1, 0]]  # neighbours swapped
            t3 = (np.arange(n)[:, None] // np.array([1, 7, 49, 343])) % V  # counting patterns
            k = np.concatenate([keys(x, push) for x in (t, t2, t3)])
            tup = np.concatenate([t, t2, t3])
            _, idx = np.unique(tup, axis=0, return_index=True)
            k = k[idx]
            print("%-28s %d distinct synthetic 4-tuples -> %d distinct keys" % ("", len(k), len(np.unique(k))))


if __name__ == "__main__":
    np.seterr(over="ignore")
    main()
