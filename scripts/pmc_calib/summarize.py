"""FETCH_SIZE / WRITE_SIZE (KB, rocprofv3) of the calibration kernels against their true byte counts -> the factor per access class."""
import csv, glob, os, sys
TRUE = {'calib_read<HIP_vector_type<unsigned int, 4u>': (1 << 30, 0), 'calib_read<HIP_vector_type<unsigned int, 2u>': (1 << 30, 0),
        'calib_read<unsigned int>': (1 << 30, 0), 'calib_read12': ((1 << 30) // 12 * 12, 0), 'calib_read16_of_64': ((1 << 30) // 4, 0),
        'calib_write<HIP_vector_type<unsigned int, 4u>': (0, 1 << 30), 'calib_write<HIP_vector_type<unsigned int, 2u>': (0, 1 << 30),
        'calib_write<unsigned int>': (0, 1 << 30), 'calib_copy16': (1 << 30, 1 << 30), 'calib_read_tiles': (1 << 30, 0)}
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            acc.setdefault((r['Kernel_Name'], r['Counter_Name']), []).append(float(r['Counter_Value']))
print('%-52s %14s %14s %10s %10s' % ('kernel (1 GiB touched once)', 'FETCH_SIZE KB', 'WRITE_SIZE KB', 'read x', 'write x'))
for key, (rd, wr) in TRUE.items():
    names = sorted({k for k, _ in acc if key in k})
    for nm in names:
        f = acc.get((nm, 'FETCH_SIZE'), [0.0]); w = acc.get((nm, 'WRITE_SIZE'), [0.0])
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        print('%-52s %14.0f %14.0f %10s %10s' % (key[:52], fm, wm, ('%.3f' % (rd / (fm * 1024))) if rd and fm else '-',
                                                  ('%.3f' % (wr / (wm * 1024))) if wr and wm else '-'))
print('read x / write x = true bytes / (counter * 1024): the factor the counter must be multiplied with for that access class')
