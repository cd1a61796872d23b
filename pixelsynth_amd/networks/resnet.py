"""ResNet-18 scene classifier of the sample ranking: the reference builds
`torchvision.models.resnet18(num_classes=365)` (models/z_buffermodel.py:88) and loads the Places365 weights
(demo.py:233-243); torchvision is not part of this build, so the same network is defined here with torchvision's module names --
a Places365 `state_dict` (after the reference's `module.` stripping) loads strictly."""
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + idt)


class ResNet18(nn.Module):
    def __init__(self, num_classes=365):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        widths, cin = (64, 128, 256, 512), 64
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(BasicBlock(cin, w, stride), BasicBlock(w, w, 1)))
            cin = w
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(F.adaptive_avg_pool2d(x, 1).flatten(1))


def resnet18(num_classes=365):
    return ResNet18(num_classes)
