"""Argument lists of the golden cases (shared by make_golden.py's harness and the GPU tests; no reference import)."""
BOX_TTP = ['--geometry', 'box', '--dimensions', '200', '200', '200',
           '--subvolumes', 'slice', '20', '0',
           '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5',
           '--bound_cond', 'T', 'T', 'P',
           '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
           '--bound_values', '302', '298']

BOX_TTRRP = ['--geometry', 'box', '--dimensions', '200', '200', '200',
             '--subvolumes', 'slice', '20', '0',
             '--bound_pos', 'relative', '-0.1', '0.5', '0.5', '1.1', '0.5', '0.5',
             '0.5', '0.5', '-0.1', '0.5', '0.5', '1.1',
             '--bound_cond', 'T', 'T', 'R', 'R', 'P',
             '--connect_pos', 'relative', '0.5', '-0.1', '0.5', '0.5', '1.1', '0.5',
             '--bound_values', '302', '298', '5', '5']

# BASELINE config 3 in small: cross-plane film, 2000 A thick, 500 x 500 A periodic cell
FILM_TTP = ['--geometry', 'box', '--dimensions', '2000', '500', '500',
            '--subvolumes', 'slice', '20', '0',
            '--bound_pos', 'relative', '0', '.5', '.5', '1', '.5', '.5',
            '--bound_cond', 'T', 'T', 'P',
            '--connect_pos', 'relative', '.5', '0', '.5', '.5', '1', '.5', '.5', '.5', '0', '.5', '.5', '1',
            '--bound_values', '302', '298']

# BASELINE config 4 in small: wire along z with 100 sides (400 triangles: the engine keeps such a mesh in global memory and
# walks the face tree), caps at 302 / 298 K, rough side wall (eta = 5 A)
WIRE = ['--geometry', 'cylinder', '--dimensions', '600', '100', '100',
        '--subvolumes', 'slice', '20', '2',
        '--bound_pos', 'relative', '0.5', '0.5', '0', '0.5', '0.5', '1',
        '--bound_cond', 'T', 'T', 'R',
        '--bound_values', '302', '298', '5']

COMMON = ['--poscar_file', 'POSCAR', '--hdf_file', 'synthetic',
          '--reference_temp', 'local', '--temp_dist', 'cold', '--temp_interp', 'linear',
          '--part_dist', 'random_subvol', '--timestep', '1', '--n_mean', '10',
          '--conv_crit', '0', '10', '--colormap', 'jet', '--fig_plot', 'energy',
          '--output', 'screen', '--max_sim_time', '0-00:00:00', '--energy_normal', 'mean']


def argv_for(case, particles, iterations=1000, extra=()):
    base = {'ttp': BOX_TTP, 'ttrrp': BOX_TTRRP, 'film': FILM_TTP, 'wire': WIRE}[case]
    return list(base) + list(COMMON) + ['--particles', 'total', str(particles),
                                        '--iterations', str(iterations)] + list(extra)
