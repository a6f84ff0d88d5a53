"""RasterTiler: tile a request into independent sub-requests (the data-parallel fan-out,
API of the reference's raster/parallelize.py:13-125).  Tiles are pasted into the mosaic
on the device (gm_paste_tile), with the reference's y-axis flip."""
import ctypes
from itertools import product

import numpy as np

from ..hip import runtime
from ..hip.array import DeviceArray, as_device
from ..hip.dtypes import code_of
from .base import BaseSingle

__all__ = ["RasterTiler"]


class RasterTiler(BaseSingle):
    """Parallelize operations on a RasterBlock by tiling the request.

    Args:
      source (RasterBlock): The source RasterBlock
      tile_size (int or list): maximum tile size in pixels, or [width, height]
    """

    def __init__(self, source, tile_size):
        if hasattr(tile_size, "__iter__"):
            if len(tile_size) != 2:
                raise ValueError("'tile_size' should be a scalar or a list of length 2.")
            tile_size = [int(x) for x in tile_size]
        else:
            tile_size = [int(tile_size), int(tile_size)]
        if tile_size[0] <= 0 or tile_size[1] <= 0:
            raise ValueError("'tile_size' should be greater than 0")
        super().__init__(source, tile_size)

    @property
    def tile_size(self):
        return self.args[1]

    def get_sources_and_requests(self, **request):
        if request["mode"] != "vals":
            return [(None, None), (self.store, request)]
        x1, y1, x2, y2 = request["bbox"]
        cell_x = (x2 - x1) / request["width"]
        cell_y = (y2 - y1) / request["height"]
        if cell_x == 0 and cell_y == 0:  # point request
            return [(None, None), (self.store, request)]
        xs = np.arange(x1, x2, cell_x * self.tile_size[0])
        ys = np.arange(y1, y2, cell_y * self.tile_size[1])
        if xs[-1] != x2:  # leftover tiles
            xs = np.append(xs, x2)
        if ys[-1] != y2:
            ys = np.append(ys, y2)
        count_x, count_y = len(xs) - 1, len(ys) - 1
        result = [({"dtype": self.dtype, "fillvalue": self.fillvalue,
                    "shape_yx": (request["height"], request["width"]),
                    "count_xy": (count_x, count_y), "tilesize_xy": self.tile_size}, None)]
        # x-major, low y first (parallelize.py:82): the order process() relies on
        for i, j in product(range(count_x), range(count_y)):
            result.append((self.store, {**request, "bbox": (xs[i], ys[j], xs[i + 1], ys[j + 1]),
                                        "width": int((xs[i + 1] - xs[i]) / cell_x),
                                        "height": int((ys[j + 1] - ys[j]) / cell_y)}))
        return result

    @staticmethod
    def process(process_kwargs, *all_data):
        if len(all_data) == 0:
            return None
        if process_kwargs is None:
            return all_data[0]  # non-tiled / meta / time requests
        first = next((d for d in all_data if d is not None), None)
        if first is None:
            return None
        runtime.require_device()
        height, width = process_kwargs["shape_yx"]
        bands = first["values"].shape[0]
        dtype = np.dtype(process_kwargs["dtype"])
        fill = process_kwargs["fillvalue"]
        mosaic = DeviceArray.empty((bands, height, width), dtype)
        L = runtime.lib()
        count_x, count_y = process_kwargs["count_xy"]
        tile_x, tile_y = process_kwargs["tilesize_xy"]
        # np.full(shape, fill) of the reference (parallelize.py:108-112) -- skipped when the tiles, which sit in disjoint
        # slots, cover every cell of the mosaic between them (a pass over the whole result that nothing would read:
        # 170 us for a 16384^2 float32 mosaic); a tile that is None or short of its slot leaves fill values behind
        covered = 0
        for (ix, iy), data in zip(product(range(count_x), range(count_y)), all_data):
            if data is not None and len(data["values"].shape) == 3 and data["values"].shape[0] == bands:
                _, th, tw = data["values"].shape
                if height - (iy * tile_y + th) >= 0 and ix * tile_x + tw <= width and th <= tile_y and tw <= tile_x:
                    covered += th * tw
        if covered != height * width or len(all_data) != count_x * count_y:
            # (a boolean view has no fill value: np.full(shape, None, dtype=bool) of the reference is all False)
            runtime.check(L.gm_fill(ctypes.c_void_p(mosaic.ptr), code_of(dtype),
                                    runtime.scalar_bits(False if fill is None else fill, dtype),
                                    mosaic.size), "gm_fill")
        tiles, th_, tw_, row_, col_ = [], [], [], [], []
        for (ix, iy), data in zip(product(range(count_x), range(count_y)), all_data):
            if data is None:
                continue
            tile = as_device(data["values"])
            if tile.dtype != dtype:
                # values[...] = tile of parallelize.py:118-124 casts on assignment (a TemporalSum below the tiler hands
                # int64 sums to a mosaic of its store's declared dtype): NumPy's astype, as a COPY program
                from .stencil import _cast_tile
                tile = _cast_tile(tile, dtype)
            if tile.shape[0] != bands:
                if tile.shape[0] != 1:
                    raise ValueError("could not broadcast input array from shape %s into shape %s"
                                     % (tuple(tile.shape), (bands,) + tuple(tile.shape[1:])))
                # values[:, rows, columns] = tile of one band: NumPy's assignment repeats it in every band of the mosaic
                from ..hip.program import _broadcast_operand
                tile = _broadcast_operand(tile, (bands,) + tuple(tile.shape[1:]))
            _, th, tw = tile.shape
            tiles.append(tile)
            th_.append(th)
            tw_.append(tw)
            row_.append(height - (iy * tile_y + th))  # low y maps to high row index (parallelize.py:114-124)
            col_.append(ix * tile_x)
        if tiles:
            n = len(tiles)
            ints = ctypes.c_int * n
            L.gm_paste_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_int]
            L.gm_paste_tiles.restype = ctypes.c_int
            runtime.check(L.gm_paste_tiles(ctypes.c_void_p(mosaic.ptr), height, width, n,
                                           (ctypes.c_void_p * n)(*[t.ptr for t in tiles]), ints(*th_), ints(*tw_),
                                           ints(*row_), ints(*col_), bands, dtype.itemsize), "gm_paste_tiles")
        return {"values": mosaic, "no_data_value": fill}

    process.__func__.gm_tiler = True  # lets the fusion pass write batched tiles straight into the mosaic
