#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference checkout (run in the build container only).

Fixtures are DATA the reference's own tests hold (test_files/*), never reference source:
  * o157_plasmid.fasta, viruses.fna, all_ns.fa, empty_fasta.fa, test.fasta, e.coli-W.fasta.gz
    -> byte copies of /root/reference/test_files/<name>
  * o157_sketch.npz -> compact re-encoding of test_files/e.coli-o157.fasta.sketch
    (pre-0.3 bincode layout): sorted (seed,pos,ctg<<1|canon) u32 triples,
    sorted u64 markers, contig lengths, contig names, c, k, total length.
  * pinned.json -> end-to-end triples printed in test_results_versions/0.3.0:130-135
    (search --median, learned ANI off) and the range assertions of tests/tests.rs,
    tests/int_test_new.rs.

/root/reference does not exist on the GPU box, hence the committed copies.
"""
import json, os, shutil, struct, sys
import numpy as np

REF = "/root/reference/test_files"
OUT = os.path.dirname(os.path.abspath(__file__))


def decode_old_sketch(path):
    b = open(path, "rb").read()
    o = 0

    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", b, o)[0]; o += 8; return v

    def u32():
        nonlocal o
        v = struct.unpack_from("<I", b, o)[0]; o += 4; return v

    def u8():
        nonlocal o
        v = b[o]; o += 1; return v

    def string():
        nonlocal o
        n = u64(); s = b[o:o + n].decode(); o += n; return s

    c, k, marker_c = u64(), u64(), u64()
    u8(); u8()                       # use_syncs, use_aa
    n = u64(); o += 8 * n            # acgt_to_aa_encoding
    n = u64(); o += n                # acgt_to_aa_letters
    u64()                            # orf_size
    file_name = string()
    assert u8() == 1                 # Option tag
    nkeys = u64()
    seeds, poss, cc = [], [], []
    for _ in range(nkeys):
        seed = u32(); m = u64()
        for _ in range(m):
            pos = u32(); canon = u8(); ctg = u32(); u8()  # phase
            seeds.append(seed); poss.append(pos); cc.append((ctg << 1) | canon)
    nct = u64(); contigs = [string() for _ in range(nct)]
    total_len = u64()
    ncl = u64(); ctg_len = [u32() for _ in range(ncl)]
    u64()                            # repetitive_kmers
    nm = u64(); markers = [u64() for _ in range(nm)]
    sk_marker_c, sk_c, sk_k, contig_order = u64(), u64(), u64(), u64()
    u8()                             # amino_acid
    assert o == len(b), (o, len(b))
    assert (sk_c, sk_k) == (c, k)
    seeds = np.array(seeds, np.uint32); poss = np.array(poss, np.uint32); cc = np.array(cc, np.uint32)
    order = np.lexsort((poss, cc >> 1, seeds))
    return dict(c=c, k=k, marker_c=marker_c, file_name=file_name, nkeys=nkeys,
                seed=seeds[order], pos=poss[order], ctgcanon=cc[order],
                markers=np.sort(np.array(markers, np.uint64)),
                contig_lengths=np.array(ctg_len, np.uint32), contigs=contigs,
                total_len=total_len)


def main():
    for name in ["o157_plasmid.fasta", "viruses.fna", "all_ns.fa", "empty_fasta.fa", "test.fasta",
                 "e.coli-W.fasta.gz"]:
        shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, name))
        os.chmod(os.path.join(OUT, name), 0o644)
    sk = decode_old_sketch(os.path.join(REF, "e.coli-o157.fasta.sketch"))
    print("o157 sketch: keys", sk["nkeys"], "positions", len(sk["seed"]), "markers", len(sk["markers"]),
          "contigs", sk["contig_lengths"], "c,k", sk["c"], sk["k"])
    np.savez_compressed(os.path.join(OUT, "o157_sketch.npz"),
                        seed=sk["seed"], pos=sk["pos"], ctgcanon=sk["ctgcanon"], markers=sk["markers"],
                        contig_lengths=sk["contig_lengths"], contigs=np.array(sk["contigs"]),
                        c=np.uint32(sk["c"]), k=np.uint32(sk["k"]), marker_c=np.uint32(sk["marker_c"]),
                        total_len=np.uint64(sk["total_len"]), file_name=np.array(sk["file_name"]))
    pinned = {
        "_source": "reference test_results_versions/0.3.0:130-135 (search --median -n 5, learned ANI off); "
                   "tests/tests.rs:42-60,130-157; tests/int_test_new.rs:57-62",
        "triples_median_percent": [
            {"ref": "o157_plasmid.fasta", "query": "o157_sketch", "ani": 100.00, "af_ref": 99.84, "af_query": 1.68},
            {"ref": "e.coli-W.fasta.gz", "query": "o157_sketch", "ani": 98.39, "af_ref": 85.46, "af_query": 75.97},
            {"ref": "o157_sketch", "query": "o157_sketch", "ani": 100.00, "af_ref": 100.00, "af_query": 100.00},
        ],
        "w_vs_w": {"ani_min": 1.0, "af_min": 0.99},
        "viruses_triangle_i": {"one_ani_in": [99.0, 99.9], "another_ani_gt": 99.9},
        "avx2_vs_scalar_120bp": {
            "seq": "ATCAGATTTAAAAAAAAATTTTGCTAGCTGATCGATCGATCGATGTGTATATATTAAAAGAGAGAGAGGGGGGGGAAAAAAAAAAAAACTGATCGATCGATGCTAGCTAGTCAGTCGATG",
            "c": 10},
        "all_n_150bp": {"seq": "N" * 149 + "n", "c": 30, "n_seeds": 0},
    }
    json.dump(pinned, open(os.path.join(OUT, "pinned.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
