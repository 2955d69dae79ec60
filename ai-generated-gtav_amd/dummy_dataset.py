"""Synthetic input fixture (reference dummy_dataset.py:6-72, web_dataset.py:22-38): 5 constant-colour frames
blue -> red (last one green when the last action is 1) and 25-way one-hot actions."""
import torch


def actions_to_one_hot(actions):
    """web_dataset.py:22-38: -1 -> zero row."""
    a = torch.as_tensor(actions)
    out = torch.zeros(len(a), 25, dtype=torch.long)
    m = a >= 0
    if m.any():
        out[torch.arange(len(a))[m], a[m]] = 1
    return out.float()


class ImageDataset(torch.utils.data.Dataset):
    def __init__(self, split: str, return_actions: bool = False, height: int = 360, width: int = 640):
        self.split, self.return_actions = split, return_actions
        blue, red, green = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([1.0, 0.0, 0.0]), torch.tensor([0.0, 1.0, 0.0])
        frames = [((1 - s) * blue + s * red).view(3, 1, 1).expand(3, height, width) for s in torch.linspace(0, 1, 5)]
        self.sequence_blue_red = torch.stack(frames)
        self.sequence_blue_green = self.sequence_blue_red.clone()
        self.sequence_blue_green[-1] = green.view(3, 1, 1).expand(3, height, width)

    def __len__(self):
        return 10000000 if self.split == "train" else 10

    def __getitem__(self, index):
        if not self.return_actions:
            return {"video": self.sequence_blue_red}
        actions = torch.randint(0, 2, (5,), dtype=torch.long)
        actions[:-1] = -1
        last = int(actions[-1])
        return {"video": self.sequence_blue_red if last == 0 else self.sequence_blue_green,
                "actions": actions_to_one_hot(list(actions))}
