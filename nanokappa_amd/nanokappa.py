"""Driver with the flow of the reference's nanokappa.py (:26-107): parameter file -> Geometry -> Phonon ->
Population -> timestep loop -> final state.

    python -m nanokappa_amd.nanokappa -ff parameters.txt
"""
import os
import re
import sys
from datetime import datetime, timedelta

from .argument_parser import read_args, generate_results_folder
from .geometry import Geometry
from .phonon import Phonon
from .population import Population


def main(argv=None):
    args = read_args(False, argv)
    args = generate_results_folder(args)
    out = None
    # nanokappa.py:34-36 compares the parsed value with 'file'; an explicit `--output file` arrives as the list ['file'] and
    # goes to the screen there (only the default writes output.txt).  Fixed here: both spellings write the file.
    mode = args.output[0] if isinstance(args.output, (list, tuple)) else args.output
    if mode == 'file':
        out = open(os.path.join(args.results_folder, 'output.txt'), 'a')
        sys.stdout = out
    with open(os.path.join(args.results_folder, 'arguments.txt'), 'w') as f:   # nanokappa.py:38-50
        for key, val in vars(args).items():
            f.write('--%s %s\n' % (key, val if isinstance(val, str) else ' '.join(str(i) for i in val)))
    mt = [int(i) for i in re.split('-|:', args.max_sim_time[0])]
    max_time = timedelta(days=mt[0], hours=mt[1], minutes=mt[2], seconds=mt[3])
    start = datetime.now()
    print('Simulation name: %s' % args.results_folder)
    geo = Geometry(args)
    phonons = Phonon(args, 0)
    pop = Population(args, geo, phonons)
    flag = True
    while flag:
        # batches end on the 100-step bookkeeping boundary, so stop criteria are checked as often as the
        # reference's residue is updated
        n = min(100 - pop.current_timestep % 100, args.iterations[0] - pop.current_timestep)
        pop.run(max(n, 1), geo, phonons)
        flag = (pop.current_timestep < args.iterations[0]) and not pop.finish_sim
        if max_time.total_seconds() > 0:
            flag = flag and datetime.now() - start < max_time
    print('Saving end of run particle data...')
    pop.write_final_state(geo)
    pop.view.postprocess()
    total = datetime.now() - start
    print('Total time: %s' % total)
    if out is not None:
        sys.stdout = sys.__stdout__
        out.close()
    return pop


if __name__ == '__main__':
    main()
