"""Parameter-file / command-line front end with the reference's flags and semantics
(reference argument_parser.py:6-181): `--from_file <txt>` splits the file on whitespace and feeds argparse.
Two additions: `--seed` (the reference uses the unseeded global NumPy generator) and `--device`."""
import argparse
import os
import sys


def initialise_parser(debug_flag=False):
    p = argparse.ArgumentParser()
    a = p.add_argument
    a('--from_file', '-ff', default='', type=str, nargs=1)
    a('--geometry', '-g', default=['cuboid'], type=str, nargs=1)
    a('--dimensions', '-d', default=[10e3, 1e3, 1e3], type=float, nargs='*')
    a('--scale', '-s', default=[1, 1, 1], type=float, nargs=3)
    a('--geo_rotation', '-gr', default=[0, 0, 0, 'xyz'], nargs='*')
    a('--mat_rotation', '-mr', default=[], nargs='*')
    a('--isotope_scat', '-is', default=[], type=int, nargs='*')
    a('--particles', '-p', default=['pmps', 1], nargs=2)
    a('--timestep', '-ts', default=[1], type=float, nargs=1)
    a('--iterations', '-i', default=[10000], type=int, nargs=1)
    a('--max_sim_time', '-mt', default=['1-00:00:00'], type=str, nargs=1)
    a('--subvolumes', '-sv', default=[], nargs='*')
    a('--temp_dist', '-td', default=['cold'], choices=['cold', 'hot', 'linear', 'mean', 'random', 'custom'], type=str, nargs='*')
    a('--temp_interp', '-ti', default=['nearest'], choices=['nearest', 'linear', 'radial'], type=str, nargs=1)
    a('--subvol_temp', '-st', default=[], type=float, nargs='*')
    a('--bound_cond', '-bc', default=[], choices=['T', 'P', 'R'], type=str, nargs='*')
    a('--bound_pos', '-bp', default=[], nargs='*')
    a('--bound_values', '-bv', default=[], type=float, nargs='*')
    a('--connect_pos', '-cp', default=[], nargs='*')
    a('--fig_plot', '-fp', default=[], type=str, nargs='*')
    a('--colormap', '-cm', default=['jet'], type=str, nargs=1)
    a('--theme', '-th', default=['white'], choices=['white', 'light', 'dark'], type=str, nargs=1)
    a('--n_mean', '-nm', default=[100], type=int, nargs=1)
    a('--conv_crit', '-cc', default=[0, 1], type=float, nargs=2)
    a('--mat_folder', '-mf', default=[''], type=str, nargs='*')
    a('--poscar_file', '-pf', required=True, type=str, nargs='*')
    a('--hdf_file', '-hf', required=True, type=str, nargs='*')
    a('--results_folder', '-rf', default=[], type=str, nargs='*')
    a('--part_dist', '-pd', default=['random_subvol'], type=str, nargs=1)
    a('--empty_subvols', '-es', default=[], type=int, nargs='*')
    a('--subvol_material', '-sm', default=[], type=int, nargs='*')
    a('--reference_temp', '-rt', default=['local'], nargs=1)
    a('--reservoir_gen', '-gn', default=['constant'], choices=['fixed_rate', 'one_to_one', 'constant'], type=str, nargs='*')
    a('--path_points', '-pp', default=[], nargs='*')
    a('--energy_normal', '-en', default=['mean'], type=str, nargs=1)
    a('--bound_scat', '-bs', default=['velocity'], type=str, nargs='*')
    a('--output', '-op', default='file', type=str, nargs=1)
    # additions of this build
    a('--seed', default=[0], type=int, nargs=1, help='seed of the counter-based RNG (Philox4x32-10)')
    a('--device', default=[0], type=int, nargs=1, help='HIP device index')
    return p


def read_args(debug_flag=False, argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if '-ff' in argv or '--from_file' in argv:
        key = '-ff' if '-ff' in argv else '--from_file'
        filename = argv[argv.index(key) + 1]
        with open(filename, 'r') as f:
            args = initialise_parser(debug_flag).parse_args(f.read().split())
        args.from_file = filename
        return args
    return initialise_parser(debug_flag).parse_args(argv)


def get_folder_index(loc):
    base, dirname = os.path.basename(loc), os.path.dirname(loc)
    if not os.path.exists(dirname):
        return 0
    same = []
    for d in os.listdir(dirname):
        if base in d:
            try:
                same.append(int(d.split('_')[-1]))
            except ValueError:
                pass
    return max(same) + 1 if same else 0


def generate_results_folder(args):
    if len(args.results_folder) == 0:
        args.results_folder = os.getcwd()
        return args
    loc = os.path.normpath(os.path.relpath(args.results_folder[0]))
    if not os.path.isabs(loc):
        loc = os.path.join(os.getcwd(), loc)
    i = get_folder_index(loc)
    os.makedirs('%s_%d' % (loc, i), exist_ok=False)
    args.results_folder = '%s_%d' % (loc, i)
    return args
