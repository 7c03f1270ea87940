#!/usr/bin/env python3
"""Merge the three tools/pmc_summary.py outputs (FETCH_SIZE pass, WRITE_SIZE pass, SQ/GRBM pass) into
profiles/pmc_latest.json (read by bench.py as roofline.traffic) and a markdown table on stdout.

    python tools/pmc_report.py fetch.json write.json sq.json out.json
Counter units as the guide prescribes: FETCH_SIZE / WRITE_SIZE count KB (x 1024 -> bytes, raw, no further correction);
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)."""
import json
import sys


def main():
    fe, wr, sq = (json.load(open(p)) for p in sys.argv[1:4])
    geometry = sys.argv[5] if len(sys.argv) > 5 else 'bench.py --minutes 20 --steps 1 --warmup 0'
    per = {}
    for k in sorted(set(fe) | set(wr) | set(sq)):
        d = {}
        src = sq.get(k) or fe.get(k) or wr.get(k)
        d['launches'] = src['launches']
        d['avg_duration_us'] = src['avg_duration_us']
        if k in fe and 'FETCH_SIZE' in fe[k]:
            d['FETCH_SIZE_bytes'] = fe[k]['FETCH_SIZE'] * 1024
        if k in wr and 'WRITE_SIZE' in wr[k]:
            d['WRITE_SIZE_bytes'] = wr[k]['WRITE_SIZE'] * 1024
        s = sq.get(k, {})
        if 'GRBM_GUI_ACTIVE' in s and s['GRBM_GUI_ACTIVE'] > 0:
            cyc = s['GRBM_GUI_ACTIVE'] / 8.0
            d['clock_ghz'] = cyc / (s['avg_duration_us'] * 1e3)
            d['mfma_busy_frac'] = s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 256 * 4)
            if s.get('SQ_LDS_IDX_ACTIVE'):
                d['lds_conflict_frac'] = s.get('SQ_LDS_BANK_CONFLICT', 0.0) / s['SQ_LDS_IDX_ACTIVE']
            if s.get('SQ_WAVE_CYCLES'):
                d['wait_inst_frac'] = s.get('SQ_WAIT_INST_ANY', 0.0) / s['SQ_WAVE_CYCLES']
                d['active_inst_frac'] = s.get('SQ_ACTIVE_INST_ANY', 0.0) / s['SQ_WAVE_CYCLES']
        per[k] = d
    # bf16x3 mode: the conv_x3* kernels (conv_igemm_kernel launches belong to the exact-f32 companion step of bench.py)
    conv = [v for k, v in per.items() if k.startswith(('conv_x3', 'issk::conv_x3', 'conv1_patch'))
            and 'FETCH_SIZE_bytes' in v and 'WRITE_SIZE_bytes' in v]
    n = sum(v['launches'] for v in conv)
    traffic = sum(v['launches'] * (v['FETCH_SIZE_bytes'] + v['WRITE_SIZE_bytes']) for v in conv) / max(n, 1)
    # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read (128-byte requests
    # tallied at 64 B) -- the conv kernels read 16 bytes per lane -- so the corrected figure doubles the fetch part
    corrected = sum(v['launches'] * (2 * v['FETCH_SIZE_bytes'] + v['WRITE_SIZE_bytes']) for v in conv) / max(n, 1)
    out = {'conv_hbm_bytes_per_launch_bf16x3': traffic, 'conv_hbm_bytes_per_launch_corrected_bf16x3': corrected,
           'source': geometry,
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on ' + geometry + ': '
                   'launch-weighted mean of FETCH_SIZE+WRITE_SIZE over the conv GEMM launches, RAW counter bytes (KB x 1024)',
           'per_kernel': per}
    # the bench line the PMC run itself printed (argv[6]): algorithmic flops per launch of every kernel instantiation in THAT run, so that
    # bench.py can scale the per-launch bytes to the mean launch of another recording length
    if len(sys.argv) > 6:
        try:
            line = [json.loads(l) for l in open(sys.argv[6]) if l.startswith('{')][-1]
            out['pmc_run_flops_per_launch'] = {k['kernel'].replace(' ', ''): k['flops_per_launch'] for k in line['roofline']['kernels']}
            out['pmc_run_minutes'] = line['config']['audio_hours_per_step_per_gpu'] * 60.0
        except Exception as exc:                               # noqa: BLE001
            out['pmc_run_flops_per_launch_error'] = repr(exc)
    json.dump(out, open(sys.argv[4], 'w'), indent=1)
    print('| kernel | launches | avg us | FETCH MB | WRITE MB | (F+W)/t TB/s | MFMA busy % | clock GHz | LDS conflict % |')
    print('|---|---|---|---|---|---|---|---|---|')
    for k, v in per.items():
        f, w = v.get('FETCH_SIZE_bytes', 0.0), v.get('WRITE_SIZE_bytes', 0.0)
        print(f"| `{k}` | {v['launches']} | {v['avg_duration_us']:.1f} | {f / 1e6:.1f} | {w / 1e6:.1f} | "
              f"{(f + w) / (v['avg_duration_us'] * 1e-6) / 1e12:.2f} | {100 * v.get('mfma_busy_frac', 0):.1f} | "
              f"{v.get('clock_ghz', 0):.2f} | {100 * v.get('lds_conflict_frac', 0):.0f} |")
    print(f'\nlaunch-weighted mean HBM traffic per conv GEMM launch: {traffic / 1e9:.3f} GB raw, {corrected / 1e9:.3f} GB with the gfx950 FETCH_SIZE x 2 correction')


if __name__ == '__main__':
    main()
