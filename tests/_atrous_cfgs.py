"""The fixture configurations of tests/golden/atrous.npz (same table as oracle/gen_golden.py:ATROUS_CFGS, which wrote it from the reference)."""
ATROUS_CFGS = {
    "atrous": ("resunet", dict(channels=1, hidden=[16, 32, 64], scale=4, depth=1, dilations=[[1, 3], [1, 2], [1]]), 32, 2),
    "psp": ("resunet", dict(channels=1, hidden=[48, 96], scale=4, depth=1, pool_sizes=[1, 2, 4], encoder_pool=False), 24, 2),
    "psp_enc": ("resunet", dict(channels=[3, 1], hidden=[32, 64], scale=2, depth=0, pool_sizes=[1, 2], encoder_pool=True), 16, 3),
    "atrous_psp": ("resunet", dict(channels=1, hidden=[32, 64], scale=4, depth=2, dilations=[[1, 5], [2]], pool_sizes=[1, 2], encoder_pool=True), 32, 1),
    "rd_atrous_psp": ("rdresunet", dict(channels=1, hidden=[32, 64], scale=2, depth=1, dilations=[[1], [1, 3]], pool_sizes=[1, 2], encoder_pool=True,
                                        rdnet_init=16, growth_rates=[8, 8, 16], ds_blocks=[False, False, True], ese_blocks=[True, False, True],
                                        n_blocks=[1, 2, 1]), 32, 2),
}
