"""sketch-reconstruction (metrics/samples.py of the reference): 18 seeded validation sketches next to their greedy
reconstructions.  The reference renders an SVG grid through svgwrite / svglib (not available here); this one draws
the same interlaced grid (original, reconstruction, ...; 6 per row) with matplotlib and returns the PNG path."""
import numpy as np

from ..core.metrics import ImageMetric


def strokes_to_lines(stroke3):
    """stroke-3 offsets -> list of (n,2) absolute polylines (pen lifts split them)."""
    s = np.asarray(stroke3, dtype=np.float64)
    xy = np.cumsum(s[:, :2], axis=0)
    lines, start = [], 0
    for i, pen in enumerate(s[:, 2]):
        if pen == 1:
            lines.append(xy[start:i + 1])
            start = i + 1
    if start < len(xy):
        lines.append(xy[start:])
    return [ln for ln in lines if len(ln)]


def stroke5_to_stroke3(rows):
    """(L,5) rows (dx, dy, p_down, p_up, p_end) -> stroke-3 up to the first end-of-sketch row."""
    rows = np.asarray(rows, dtype=np.float64)
    end = np.nonzero(rows[:, 2:].argmax(-1) == 2)[0]
    n = int(end[0]) if len(end) else len(rows)
    return np.c_[rows[:n, :2], (rows[:n, 2:].argmax(-1) == 1).astype(np.float64)]


class ReconstructedSketchSamples(ImageMetric):
    name = 'sketch-reconstruction'
    input_type = 'predictions_on_validation_set'

    def compute(self, input_data):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        x, y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, is_continuous = input_data
        np.random.seed(19)
        idx = np.random.permutation(len(x))[:18]
        np.random.seed()
        pairs = []
        for i in idx:
            if is_continuous:
                a, b = stroke5_to_stroke3(x[i]), stroke5_to_stroke3(pred_x[i][1:])     # row 0 of a reconstruction = start symbol
            else:
                a, b = tokenizer.decode_single(x[i]), tokenizer.decode_single(pred_x[i])
            pairs += [a, b]
        cols = 6
        rows = (len(pairs) + cols - 1) // cols
        fig, axes = plt.subplots(rows, cols, figsize=(2 * cols, 2 * rows), squeeze=False)
        for k, ax in enumerate(axes.reshape(-1)):
            ax.axis("off")
            if k < len(pairs):
                for ln in strokes_to_lines(pairs[k]):
                    ax.plot(ln[:, 0], -ln[:, 1], color="k" if k % 2 == 0 else "tab:blue", linewidth=1)
                ax.set_aspect("equal")
        out = tmp_filepath.format('reconstruction')
        fig.savefig(out, dpi=60)
        plt.close(fig)
        return out
