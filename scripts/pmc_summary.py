"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel; calibrate FETCH/WRITE with k_cal_stream."""
import csv, glob, collections, json, sys

def load(d):
    f = glob.glob('%s/**/*counter_collection.csv' % d, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v[-5:]) / len(v[-5:]) for c, v in cs.items()} for k, cs in agg.items()}

if __name__ == '__main__':
    sq, fe, wr = sys.argv[1:4]
    known_r, known_w = int(sys.argv[4]), int(sys.argv[5])
    S, F, W = load(sq), load(fe), load(wr)
    fr = known_r / (F['k_cal_stream']['FETCH_SIZE'] * 1024)
    fw = known_w / (W['k_cal_stream']['WRITE_SIZE'] * 1024)
    out = {'calibration': {'read_factor': fr, 'write_factor': fw, 'true_read_bytes': known_r, 'true_write_bytes': known_w}}
    for k in S:
        if not k.startswith('k_'):
            continue
        e = dict(S[k])
        e.update({c: v for c, v in W.get(k, {}).items()})
        e['read_bytes'] = F.get(k, {}).get('FETCH_SIZE', 0) * 1024 * fr
        e['write_bytes'] = W.get(k, {}).get('WRITE_SIZE', 0) * 1024 * fw
        out[k] = e
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
