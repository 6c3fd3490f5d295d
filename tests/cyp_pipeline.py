"""diplotype_cyp2d6 (src/cyp2d6/caller.rs:39-741) assembled from the CPU oracle's pieces -- the expected result for
sp_cyp_diplotype (test infrastructure).  Steps and reference lines are those of the library's driver (pb-starphase_amd/csrc/sp_cyp_call.hip)."""
import ctypes as C

import numpy as np

import oracle_ffi as of

T = of.REGION_TYPES
SEEDS = {T["CYP2D6*5"]: 0, T["REP6"]: 1, T["REP7"]: 2, T["spacer"]: 3, T["link_region"]: 4}      # caller.rs:229-236


def score(h, penalize):
    """MappingStats::custom_score (src/data_types/mapping.rs:60-84)"""
    ln = h["seq_len"] if penalize else h["seq_len"] - h["unmapped"]
    return max(float(h["nm"] + (h["unmapped"] if penalize else 0)), 0.1) / float(ln)


def full_allele(oracle, typ, sub):
    out = C.create_string_buffer(256)
    oracle.L.osp_cyp_full_allele(int(typ), sub.encode() if sub is not None else None, out, 256)
    return out.value.decode()


def simplify(oracle, typ, sub, cfg):
    keep = []
    c = of._cyp_cfg_struct(cfg, keep)
    out = C.create_string_buffer(256)
    oracle.L.osp_cyp_simplify_allele(int(typ), sub.encode() if sub is not None else None, 1, C.byref(c), out, 256)
    return out.value.decode()


class Db:
    """the typing database a sample needs: templates in full_allele() order, backbone + variants, star-allele definitions"""
    def __init__(self, names, types, subtypes, seqs, deep, backbone, variants, is_vi, allele_subtypes, hap_matrix, var_labels=None):
        self.var_labels = var_labels
        self.names, self.types, self.subtypes, self.seqs, self.deep = names, np.asarray(types, np.int32), subtypes, seqs, deep
        self.backbone, self.variants, self.is_vi = backbone, variants, np.asarray(is_vi, np.uint8)
        self.allele_subtypes, self.hap_matrix = allele_subtypes, np.asarray(hap_matrix, np.uint8)


def score_alleles(oracle, db, states):
    nv, na = len(db.variants), len(db.allele_subtypes)
    bv, ba = C.c_uint32(0), C.c_uint32(0)
    tie = np.zeros(max(1, na), np.uint8)
    st = np.ascontiguousarray(states, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.L.osp_cyp_score_alleles(nv, na, p(np.ascontiguousarray(db.hap_matrix)), p(db.is_vi), p(st), C.byref(bv), C.byref(ba), p(tie))
    return (bv.value, ba.value), tie[:na]


def deep_tail(db, allele, states):
    """the variants of a typed sequence against its assigned star allele as Cyp2d6Region::deep_label lists them (src/cyp2d6/haplotyper.rs:546-595,
    src/cyp2d6/region.rs:60-91): +unexpected, -missing, ?ambiguous / unknown-but-expected"""
    if db.var_labels is None:
        return ""
    out = []
    for v, label in enumerate(db.var_labels):
        hv, sv = int(db.hap_matrix[allele][v]), int(states[v])
        sign = {(0, 1): "+", (0, 2): "?", (1, 0): "-", (1, 2): "?", (1, 3): "?"}.get((hv, sv))
        if sign:
            out.append(f" {sign}{label}")
    return "".join(out)


def region_variants(db, allele, states):
    """Cyp2d6Region::variants of a typed sequence (src/cyp2d6/haplotyper.rs:546-595): every variant but reference-on-reference, with its
    VariantAlleleRelationship -- the lists `cyp2d6_alleles.json` holds (src/cyp2d6/debug.rs:29-37)"""
    names = {(0, 1): "Unexpected", (0, 2): "AmbiguousUnexpected", (0, 3): "UnknownUnexpected",
             (1, 0): "Missing", (1, 1): "Match", (1, 2): "AmbiguousMissing", (1, 3): "UnknownMissing"}
    out = []
    for v, label in enumerate(db.var_labels):
        key = (int(db.hap_matrix[allele][v]), int(states[v]))
        if key in names:
            out.append({"label": label, "is_vi": bool(db.is_vi[v]), "variant_state": names[key]})
    return out


def deep_hap_string(oracle, chain, labels, tails):
    """convert_chain_to_hap at Cyp2d6DetailLevel::DeepAlleles (src/cyp2d6/caller.rs:907-957)"""
    two_d = (T["CYP2D6"], T["CYP2D7"], T["CYP2D6*5"], T["Hybrid"])
    keep = [h for h in reversed(chain) if labels[h][0] in two_d and labels[h][0] != T["CYP2D7"]]
    if any(labels[h][0] != T["CYP2D6*5"] for h in keep):
        keep = [h for h in keep if labels[h][0] != T["CYP2D6*5"]]
    parts = []
    for h in keep:
        text = f"({h}_{full_allele(oracle, *labels[h])}{tails[h]})"
        if parts and parts[-1][0] == text:
            parts[-1][1] += 1
        else:
            parts.append([text, 1])
    return " + ".join(t + (f"x{n}" if n > 1 else "") for t, n in parts)


class ContractAligner:
    """the three alignment call sites of the CYP2D6 caller on the library's alignment contract (oracle/align.c: anchor + banded unit-cost alignment).
    tests/cpu_port_cyp.py holds the same three on the minimap2 restatement, in the reference's own call pattern."""

    def find_base_type(self, oracle, seq, db, max_missing):
        return of.oracle_find_base_type(oracle, seq, db.seqs, db.types, max_missing)

    def weight_sequence(self, oracle, seq, consensus, allowed):
        return of.oracle_weight_sequence(oracle, seq, consensus, allowed)

    def variant_states(self, oracle, seq, db):
        return of.oracle_variant_states(oracle, seq, db.backbone, [v[0] for v in db.variants], [v[1] for v in db.variants], [v[2] for v in db.variants])[0]


CONTRACT = ContractAligner()


def full_type(oracle, db, seq, max_missing, force, tail=None, aligner=CONTRACT):
    """find_full_type_in_sequence + assign_haplotype (src/cyp2d6/haplotyper.rs:326-602) -> (type, subtype); tail (a list) receives the
    deep-label variant list of the sequence"""
    hits = aligner.find_base_type(oracle, seq, db, max_missing) if seq else []
    if len(hits) == 0:
        return (T["UNKNOWN"], None)                                          # "no matches found" -> Unknown (caller.rs:350-355)
    best = min(range(len(hits)), key=lambda i: (score(hits[i], True), i))
    t = int(hits[best]["template_idx"])
    if not db.deep[t]:
        return (int(db.types[t]), db.subtypes[t])
    states = aligner.variant_states(oracle, seq, db)
    (bv, ba), tie = score_alleles(oracle, db, states)
    cands = [(T["CYP2D6"], db.allele_subtypes[a]) for a in range(len(tie)) if tie[a]]
    if (bv, ba) == (0, 0):
        cands.append((T["UNKNOWN"], None))
    if not cands:
        return (T["UNKNOWN"], None)
    if len(cands) > 1:
        cands.sort(key=lambda c: full_allele(oracle, *c))
        if not force:
            return (T["UNKNOWN"], None)
    if tail is not None and cands[0][0] == T["CYP2D6"] and cands[0][1] is not None:
        allele = list(db.allele_subtypes).index(cands[0][1])
        tail.append((deep_tail(db, allele, states), region_variants(db, allele, states) if db.var_labels is not None else None))
    return cands[0]


def diplotype(oracle, db, reads, cfg=None, min_count=3, min_af=0.10, delta=100, infer=False, normalize_d6_only=False, aligner=CONTRACT, regions=None, weigh=None,
              stages=None, retry_ladder=False):
    """aligner: the alignment call sites (ContractAligner, or the minimap2 port of tests/cpu_port_cyp.py); regions / weigh: the per-read region search and the
    per-segment weights when the caller has computed them elsewhere (forked workers); stages (a dict) receives the intermediate results"""
    cfg = cfg or of.default_cyp_config()
    out = dict(status=0, hap1="", hap2="")
    # 1. regions of interest (caller.rs:126-139)
    if regions is None:
        regions = [aligner.find_base_type(oracle, r, db, 0.5) for r in reads]
    if stages is not None:
        stages["regions"] = regions
    # 2. sequences for the consensus (caller.rs:176-213)
    raw, hpc, boff, hoff, seeds = [], [], [], [], []
    for r, hits in enumerate(regions):
        for h in hits:
            if score(h, True) > 0.5:
                continue
            seq = reads[r][int(h["start"]):int(h["end"])]
            clip = int(h["clip_start"])
            raw.append(seq); hpc.append(oracle.hpc(seq))
            boff.append(None if clip == 0 else clip + 50)
            hp = oracle.hpc_pos(db.seqs[int(h["template_idx"])], clip)
            hoff.append(None if hp == 0 else hp + 50)
            seeds.append(SEEDS.get(int(db.types[int(h["template_idx"])])))
    if not raw:
        out["status"] = 1                                                    # NO_READS (caller.rs:254-266)
        return out
    ccfg = of.cons_config(min_count=min_count, min_af=min_af, dual_max_ed_delta=delta, early_termination=True, dual=True, offset_window=100, offset_compare_length=100)
    if stages is not None:
        stages["consensus_inputs"] = dict(hpc=hpc, raw=raw, hoff=hoff, boff=boff, seeds=seeds)
    group_of, cons = of.oracle_priority_consensus(oracle, [hpc, raw], ccfg, [hoff, boff], seeds, retry_ladder=retry_ladder)
    # 4. merge_consensus_results (caller.rs:750-898)
    cset, uset = {}, {}
    for g, (hc, fc) in enumerate(cons):
        lab = full_type(oracle, db, fc, 0.1, False, aligner=aligner)
        if lab[0] in (T["UNKNOWN"], T["FalseAllele"]):
            uset.setdefault(hc, []).append(g)
        else:
            cset.setdefault((hc, simplify(oracle, lab[0], lab[1], cfg)), []).append(g)
    ignored = set()
    for hc in sorted(uset):
        others = [k for k in sorted(cset) if k[0] == hc]
        if len(others) == 1:
            cset[others[0]] += uset[hc]
        else:
            if len(others) > 1:
                ignored.add((hc, "UNKNOWN"))
            cset[(hc, "UNKNOWN")] = uset[hc]
    single = of.cons_config(min_count=min_count, min_af=min_af, dual_max_ed_delta=delta, early_termination=True, dual=False, offset_window=100, offset_compare_length=100)
    final, seq_idx = [], [-1] * len(raw)
    for key in sorted(cset):
        members = [s for s in range(len(raw)) if group_of[s] in cset[key]]
        for s in members:
            seq_idx[s] = len(final)
        if key in ignored:
            final.append("")
        elif len(cset[key]) == 1:
            final.append(cons[cset[key][0]][1])
        else:
            vals = [0 if boff[s] is None else boff[s] for s in members]
            mn = min(vals)
            offs = [None if v == mn else v - mn + (0 if mn == 0 else 50) for v in vals]
            final.append(of.oracle_consensus(oracle, [raw[s] for s in members], offs, single)["cons"][0])
    # 5. typing (caller.rs:331-375)
    labels, seen, tails, region_lists = [], set(), [], {}
    for fc in final:
        t = []
        lab = full_type(oracle, db, fc, 0.1, True, t, aligner=aligner)
        tails.append(t[0][0] if t else "")
        if t and t[0][1] is not None:
            region_lists[len(tails) - 1] = t[0][1]
        if fc in seen:                                                       # two groups with the same sequence: mark_false_allele keeps the subtype
            lab = (T["FalseAllele"], lab[1])
        else:
            seen.add(fc)
        labels.append(lab)
    out.update(consensus=final, labels=list(labels), sequence_indices=seq_idx)
    # 6. chains (caller.rs:429-640)
    segs, seg_off = [], [0]
    for r, hits in enumerate(regions):
        segs += [reads[r][int(h["start"]):int(h["end"])] for h in hits]
        seg_off.append(len(segs))
    allowed = np.array([lab[0] not in (T["UNKNOWN"], T["FalseAllele"]) for lab in labels], np.uint8)
    ed, ov, kept = [], [], []
    for e, o, k in (weigh(segs, final, allowed) if weigh else (aligner.weight_sequence(oracle, s, final, allowed) for s in segs)):
        ed.append(e); ov.append(o); kept.append(k)
    ed, ov, kept = np.array(ed, np.uint64).reshape(len(segs), len(final)), np.array(ov, np.float64).reshape(len(segs), len(final)), np.array(kept, np.uint8)
    if stages is not None:
        stages.update(ed=ed, ov=ov, kept=kept, seg_off=seg_off, group_of=group_of, n_inputs=len(raw), allowed=[bool(x) for x in allowed])
    types = np.array([lab[0] for lab in labels], np.int32)
    built = of.oracle_build_chains(oracle, types, np.array(seg_off, np.uint32), ed.reshape(-1), kept)
    if built is None:
        out["status"] = 7
        return out
    labels2 = [(T["FalseAllele"], s) if built["false_allele"][h] else (t, s) for h, (t, s) in enumerate(labels)]
    obs = {f"r{r:06d}": built["chains"][k] for k, r in enumerate(built["read_index"])}
    scores = {f"r{r:06d}": [[(int(ed[sg][c]), float(ov[sg][c])) for c in range(len(labels2))] for sg in built["w_rows"][k]] for k, r in enumerate(built["read_index"])}
    inp = of.ChainInputs(labels2, obs, scores, infer, not normalize_d6_only, of.DEFAULT_PENALTIES, False, cfg)
    res = of.oracle_chain_pair(oracle, inp)
    out.update(status=int(res.status), labels=labels2)
    if res.status == 0:
        ch1, ch2 = list(res.chain1[:res.n1]), list(res.chain2[:res.n2])
        out.update(chain1=ch1, chain2=ch2, score=res.score,
                   hap1=of.chain_hap_string(oracle, ch1, labels2, 1, cfg), hap2=of.chain_hap_string(oracle, ch2, labels2, 1, cfg),
                   core1=of.chain_hap_string(oracle, ch1, labels2, 0, cfg), core2=of.chain_hap_string(oracle, ch2, labels2, 0, cfg),
                   deep1=deep_hap_string(oracle, ch1, labels2, tails), deep2=deep_hap_string(oracle, ch2, labels2, tails))
        # DeeplotypeDebug (src/cyp2d6/debug.rs:10-70): cyp2d6_alleles.json
        out["alleles_json"] = {
            "hap1": {"deep_form": out["deep1"], "suballele_form": out["hap1"], "core_form": out["core1"]},
            "hap2": {"deep_form": out["deep2"], "suballele_form": out["hap2"], "core_form": out["core2"]},
            "alleles": {k: v for k, v in sorted((f"{h}_{full_allele(oracle, *labels2[h])}", lst) for h, lst in region_lists.items())}}
    return out
